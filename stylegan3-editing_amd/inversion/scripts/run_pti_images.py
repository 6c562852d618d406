"""Pivotal tuning (PTI) of the generator around one inverted image (reference inversion/scripts/run_pti_images.py:104-177).

The generator weights (everything in `synthesis` except the Fourier-feature input's three parameters) are tuned with
Adam so that synthesis(latent) reproduces the target:  loss = l2_lambda * MSE + lpips_lambda * LPIPS.  The forward is
the fp32 synthesis path on the HIP kernels; the backward runs through the ops' autograd functions (filtered_lrelu:
sign-tensor adjoint; modulated conv: gradients of the reference formulation).

LPIPS needs pretrained AlexNet weights that are not part of this package: pass `lpips_loss` (a callable
(generated, real) -> scalar tensor) or set opts.lpips_lambda = 0.  Image decoding / the side-by-side JPEG dumps of the
reference are I/O outside the hot path: targets are tensors in [-1, 1].
"""
import types

import numpy as np
import torch
from torch import nn


class _FusedAdam(torch.optim.Adam):
    """torch.optim.Adam(fused=True) -- the update of all tensors in ONE multi-tensor launch -- that still tells the rest of the
    process that the weights changed: the fused update does not advance the parameters' version counters (torch 2.10), and the
    generator's inference caches (packed convolution weights, affine packs; keyed by address and version) and the graphs'
    staleness check would keep serving the weights from before the step."""

    def step(self, closure=None):
        out = super().step(closure)
        for group in self.param_groups:
            torch.autograd.graph.increment_version(group['params'])
        return out


def tuning_optimizer(params, lr):
    """The reference's `torch.optim.Adam(params, lr=lr)` (run_pti_images.py:111-116); fused into one launch per step when every
    parameter is a float32 GPU tensor (0.4 ms of a 14 ms step at FFHQ-1024), the default implementation otherwise."""
    params = list(params)
    if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        return _FusedAdam(params, lr=lr, fused=True)
    return torch.optim.Adam(params, lr=lr)


def default_opts(**overrides):
    """Defaults of the reference RunConfig (:25-60) that the optimisation itself reads."""
    o = types.SimpleNamespace(device='cuda', steps=350, learning_rate=3e-4, lpips_lambda=1.0, l2_lambda=1.0,
                              lpips_threshold=0.06, batch_size=2, num_workers=0, save_interval=None,
                              model_save_interval=None, save_final_model=False, output_path=None)
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


def as_image_batch(t, device):
    t = torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).to(device).float()
    return t.unsqueeze(0) if t.ndim == 3 else t


class PTI:

    def __init__(self, opts, lpips_loss=None):
        self.opts = opts
        self.device = opts.device
        self.mse_loss = nn.MSELoss().to(self.device).eval()
        self.lpips_loss = lpips_loss
        if opts.lpips_lambda > 0 and lpips_loss is None:
            raise RuntimeError('PTI: opts.lpips_lambda > 0 needs an lpips_loss callable (the pretrained LPIPS network is external)')
        self.history = []                                 # (step, loss, lpips, l2) floats of the last optimize_model call

    @staticmethod
    def tunable_parameters(generator):
        """Everything in `synthesis` except the three parameters of the Fourier-feature input (its mixing weight and the affine
        that predicts the transform), which stay frozen as in the reference (:111-114)."""
        frozen = {id(p) for p in generator.synthesis.input.parameters()}
        tunable = [p for p in generator.synthesis.parameters() if id(p) not in frozen]
        assert len(tunable) == len(list(generator.synthesis.parameters())) - 3
        return tunable

    def get_optimizer(self, generator):
        """Adam(lr) over the tunable parameters, as the reference (:111-116); see `tuning_optimizer`."""
        return tuning_optimizer(self.tunable_parameters(generator), self.opts.learning_rate)

    def optimize_model(self, generator, codes, target_images, landmarks_transforms=None, image_name=None):
        optimizer = self.get_optimizer(generator)
        latents = torch.from_numpy(np.asarray(codes)).to(self.device).unsqueeze(0)
        targets = as_image_batch(target_images, self.device)
        outputs = None
        self.history = []
        if landmarks_transforms is not None:
            generator.synthesis.input.transform = torch.from_numpy(np.asarray(landmarks_transforms)).to(self.device).float()
        for step in range(self.opts.steps):
            outputs = generator.synthesis(latents, noise_mode='const', force_fp32=True)
            loss, lpips_loss, l2_loss_val = self.calc_loss(outputs, targets)
            if lpips_loss is not None and lpips_loss < self.opts.lpips_threshold:
                break
            self.history.append((step, float(loss.detach()), None if lpips_loss is None else float(lpips_loss.detach()),
                                 None if l2_loss_val is None else float(l2_loss_val.detach())))
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        if self.opts.save_final_model and self.opts.output_path is not None:
            name = (image_name or 'image').split('.')[0]
            torch.save(generator.state_dict(), self.opts.output_path / f'final_pti_model_{name}.pt')
        return outputs

    def calc_loss(self, generated_images, real_images):
        """(total, lpips term or None, l2 term or None): total = l2_lambda * MSE + lpips_lambda * LPIPS, each term only when its
        weight is positive (reference :167-177)."""
        terms = {'l2': (self.opts.l2_lambda, self.mse_loss), 'lpips': (self.opts.lpips_lambda, self.lpips_loss)}
        values = {name: (fn(generated_images, real_images) if weight > 0 else None) for name, (weight, fn) in terms.items()}
        total = sum(terms[name][0] * v for name, v in values.items() if v is not None) if any(v is not None for v in values.values()) else 0.0
        return total, values['lpips'], values['l2']

    @staticmethod
    def get_description(step, loss, lpips_loss, l2_loss_val):
        """Progress line of the reference (:179-186): 'Step: N - Loss: x[, LPIPS: y][, L2: z]'."""
        parts = [f'Step: {step} - Loss: {loss.item():.4f}']
        parts += [f'{label}: {value.item():.4f}' for label, value in (('LPIPS', lpips_loss), ('L2', l2_loss_val)) if value is not None]
        return ', '.join(parts)

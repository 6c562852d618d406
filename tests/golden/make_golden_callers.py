"""Golden fixtures for the CALLERS of the synthesis path ("caller arithmetic unpinned"), produced by
running the REFERENCE's own caller code on CPU with the reference's own Generator:

  fov/*          utils/fov_expansion.py          Expander.generate_expanded_image (9 tile transforms, merge)
  video/*        inversion/video/post_processing.py   postprocess_and_smooth_inversions (fine-layer mean, 5-tap smoothing, frames)
  styleclip/*    editing/styleclip_global_directions/{global_direction,edit}.py   get_delta_s, edit_image (55-style sweep, here 2 x 3)
  psp/*          models/setgan/encoder/psp3.py   pSp.forward (latent_avg / residual step, identity + landmark transforms, face_pool)
  restyle/*      utils/inference_utils.py        get_average_image, run_on_batch (the ReStyle loop)

Run in the build container only:   python tests/golden/make_golden_callers.py   ->  tests/golden/callers.npz

How the reference code is run, and what is NOT the reference here (all of it visible below, nothing hidden):
  * `imageio`, `clip`, `pyrallis`, `torchvision` are not installed.  The modules above only import them (video / JPEG writing,
    the CLIP text encoder, the CLI wrapper); inert placeholder modules satisfy those import statements, and none of their
    attributes is reached by the functions exercised (the CLIP text direction is an INPUT of the fixture: `get_delta_i` is
    replaced by a function returning the synthetic direction; CLIP weights do not exist offline).
  * The reference pins tensors with `.cuda()`; there is no GPU here, so `torch.Tensor.cuda` is the identity while this script
    runs, and `.to('cuda')` ignores the device (placement, not arithmetic).
  * psp3.py and inference_utils.py import module paths that do not exist in the reference tree (`inversion.models.*`),
    so they cannot be imported at all.  The two functions are taken from the files' own text at run time (ast source segment,
    executed as is, never written anywhere) -- the same code an import would have run.
  * The 186 M-parameter IR-SE50 encoder needs the CUDA-compiled StyleGAN2 ops at import (map2style -> stylegan2.model); the
    loop is driven with a small stand-in encoder (a linear map of the image means, defined in tests/callers_common.py and used
    identically by the product / oracle tests): the fixture pins the LOOP and the wrapper, the encoder has its own fixture.
Nothing from the reference is copied: the fixture holds seeded inputs' reference OUTPUTS only.
"""
import ast
import dataclasses
import os
import sys
import types
from typing import Optional  # noqa: F401  (name used by the executed reference functions' annotations)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, TESTS)
sys.path.insert(0, REF)

import torch  # noqa: E402

for name in ('imageio', 'clip', 'torchvision'):
    sys.modules.setdefault(name, types.ModuleType(name))
_pyrallis = types.ModuleType('pyrallis')
_pyrallis.wrap = lambda *a, **k: (lambda fn: fn)
sys.modules.setdefault('pyrallis', _pyrallis)
torch.Tensor.cuda = lambda self, *a, **k: self           # no GPU in the build container
_tensor_to = torch.Tensor.to


def _to_without_cuda(self, *args, **kwargs):             # `.to('cuda')` (inference_utils.py:63): placement only
    args = tuple(a for a in args if not (isinstance(a, (str, torch.device)) and str(a).startswith('cuda')))
    if str(kwargs.get('device', '')).startswith('cuda'):
        kwargs.pop('device')
    return _tensor_to(self, *args, **kwargs) if (args or kwargs) else self


torch.Tensor.to = _to_without_cuda
torch.set_grad_enabled(False)

from callers_common import TinyEncoder, landmark, restyle_case, styleclip_case, sweep_opts  # noqa: E402
from synth_weights import CONFIGS, synth_state_dict, synth_ws  # noqa: E402

from models.stylegan3.networks_stylegan3 import Generator  # noqa: E402
from utils import common  # noqa: E402
from utils.fov_expansion import Expander  # noqa: E402
from inversion.video import post_processing  # noqa: E402
from editing.styleclip_global_directions import edit as ref_edit  # noqa: E402
from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection  # noqa: E402


def ref_generator(cfg, seed=0):
    G = Generator(**CONFIGS[cfg])
    man = {k: list(v.shape) for k, v in G.state_dict().items()}
    sd = synth_state_dict(man, seed=seed, input_bandwidth=float(G.synthesis.input.bandwidth))
    missing, unexpected = G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith('_filter') for k in missing)
    return G.eval().requires_grad_(False)


def reference_function(path, name, cls=None, **namespace):
    """The function `name` (of class `cls`) exactly as the reference file defines it, from a file that cannot be imported."""
    text = open(os.path.join(REF, path)).read()
    body = ast.parse(text).body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    import textwrap
    ns = dict(namespace)
    exec(compile(textwrap.dedent(ast.get_source_segment(text, node)), path, 'exec'), ns)
    return ns[name]


def gen_fov(out):
    for cfg in ('Ttiny', 'Rtiny'):
        G = ref_generator(cfg)
        ws = torch.from_numpy(synth_ws(2, G.num_ws, G.w_dim, seed=4))
        img = Expander(G).generate_expanded_image(ws=ws, landmark_t=landmark(), pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
        out[f'fov/{cfg}/img'] = img.numpy()


def gen_video(out):
    G = ref_generator('Ttiny')
    lat = synth_ws(7, G.num_ws, G.w_dim, seed=9)
    # per-frame landmark transforms are required: without them the reference takes `get_identity_transform()` (a numpy array) and
    # calls `.cpu()` on it (post_processing.py:26-29) -- that branch cannot run
    tr = [torch.from_numpy(np.linalg.inv(common.make_transform((0.02 * i, -0.01 * i), 2.0 * i))) for i in range(7)]
    results = {'result_latents': {f'{i:04d}': lat[i] for i in range(7)}, 'landmarks_transforms': tr}
    opts = types.SimpleNamespace(expansion_amounts=[4, 2, 0, 3], landmarks_transforms_path='given')
    frames = post_processing.postprocess_and_smooth_inversions(results, types.SimpleNamespace(decoder=G), opts)
    out['video/frames'] = np.stack(frames)                                   # [3, H, W, 3] uint8
    out['video/smooth_ws'] = post_processing.smooth_ws(lat.copy())
    _, sm_t = post_processing.smooth_latents_and_transforms(lat.copy(), tr, opts)
    out['video/smooth_transforms'] = sm_t.numpy()


def gen_styleclip(out):
    for cfg in ('Ttiny', 'Rtiny'):
        G = ref_generator(cfg)
        lat = synth_ws(1, G.num_ws, G.w_dim, seed=12)[0]
        s_avg = G.synthesis.W2S(G.mapping.w_avg.unsqueeze(0).repeat(1, G.num_ws, 1))
        delta_i_c, delta_i, s_std = styleclip_case(s_avg)
        calc = StyleCLIPGlobalDirection.__new__(StyleCLIPGlobalDirection)    # __init__ loads CLIP ViT-B/32: unavailable
        calc.delta_i_c, calc.s_avg = torch.from_numpy(delta_i_c), s_avg
        calc.s_std = {k: torch.from_numpy(v) for k, v in s_std.items()}
        calc.get_delta_i = lambda prompts: torch.from_numpy(delta_i)         # the text direction is an input of the fixture
        opts = sweep_opts()
        direction = calc.get_delta_s(opts.neutral_text, opts.target_text, 0.25)
        for k, v in direction.items():
            out[f'styleclip/{cfg}/delta_s/{k}'] = v.numpy()
        results, latents = ref_edit.edit_image(lat, landmark().astype(np.float32), G, calc, opts, image_name=None, save=False)
        out[f'styleclip/{cfg}/results'] = results.numpy()
        out[f'styleclip/{cfg}/last_latent_input'] = latents[-1]['input'].numpy()


def gen_restyle(out):
    forward = reference_function('models/setgan/encoder/psp3.py', 'forward', cls='pSp', torch=torch, common=common)
    get_average_image = reference_function('utils/inference_utils.py', 'get_average_image', torch=torch)
    run_on_batch = reference_function('utils/inference_utils.py', 'run_on_batch', torch=torch, dataclasses=dataclasses, Optional=Optional,
                                      TrainOptions=object)

    @dataclasses.dataclass
    class Opts:                                # run_on_batch calls dataclasses.asdict(opts) (inference_utils.py:72)
        n_iters_per_batch: int = 3
        resize_outputs: bool = False

    ref_forward = forward

    class Net(torch.nn.Module):                # the attributes pSp.forward / run_on_batch touch (psp3.py:13-19)
        forward = ref_forward

        def __init__(self, G):
            super().__init__()
            self.encoder = TinyEncoder()
            self.decoder = G
            self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
            self.latent_avg = G.mapping.w_avg

    for cfg in ('Tmini', 'Rmini'):
        G = ref_generator(cfg)
        net = Net(G).eval()
        x, tr = restyle_case(2)
        avg = get_average_image(net)
        out[f'restyle/{cfg}/avg_image'] = avg.numpy()[:, ::4, ::4]
        # pSp.forward: first step (latent_avg), residual step, input_code, resize, both returns with landmark transforms
        x6 = torch.cat([torch.from_numpy(x), avg.unsqueeze(0).repeat(2, 1, 1, 1)], dim=1)
        img0, lat0 = forward(net, x6, latent=None, return_latents=True, resize=False)
        img1, un1, lat1 = forward(net, x6, latent=lat0, landmarks_transform=torch.from_numpy(tr), return_aligned_and_unaligned=True, resize=True)
        out[f'psp/{cfg}/img0'], out[f'psp/{cfg}/lat0'] = img0.numpy(), lat0.numpy()
        out[f'psp/{cfg}/img1'], out[f'psp/{cfg}/un1'], out[f'psp/{cfg}/lat1'] = img1.numpy()[:, :, ::4, ::4], un1.numpy()[:, :, ::4, ::4], lat1.numpy()
        for key, lt in (('off', None), ('on', torch.from_numpy(tr))):
            imgs, lats = run_on_batch(torch.from_numpy(x), net, Opts(), avg, landmarks_transform=lt)
            out[f'restyle/{cfg}/{key}/images'] = np.stack([np.stack([imgs[i][it].numpy() for i in range(2)]) for it in range(3)])
            out[f'restyle/{cfg}/{key}/latents'] = np.stack([np.stack([lats[i][it] for i in range(2)]) for it in range(3)])


def main():
    out = {}
    gen_fov(out)
    gen_video(out)
    gen_styleclip(out)
    gen_restyle(out)
    path = os.path.join(HERE, 'callers.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB')
    for k, v in out.items():
        print(f'  {k:44s} {str(v.shape):22s} {v.dtype}')


if __name__ == '__main__':
    main()

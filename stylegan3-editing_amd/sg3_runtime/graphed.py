"""hipGraph replay of `Generator.synthesis` for a fixed batch shape.

A synthesis forward is ~190 launches (15 x {prep, conv, filtered_lrelu} plus the small affine / Fourier-feature torch
ops); at batch 8 the kernels take ~24 ms and the launch gaps ~3 ms.  Capturing the whole forward once into a HIP graph
(torch.cuda.CUDAGraph: our ctypes launches go to torch's current stream, which is the capture stream) and replaying
it removes the gaps and all Python overhead.  The graph owns static input/output buffers: `ws` (or the StyleSpace dict)
is copied in, the image comes out of a static tensor that is overwritten by the next replay.

What a captured graph bakes in, and how a replay stays valid:
  * device pointers of every tensor it reads.  `synthesis.input.transform` is REBOUND by the callers (pSp.forward,
    PTI, the FOV expander assign a fresh tensor per call), so the graph owns a static [batch,3,3] transform buffer.  That
    buffer shadows the module attribute ONLY while the graph is warmed up and captured; afterwards the caller's own tensor is
    back on the module (eager calls with another batch size, `state_dict()` and `load_state_dict` see the user's [3,3] or
    [B,3,3] transform), and every replay copies whatever transform the module (or the `transform=` argument) holds then into
    the static buffer;
  * host-side values derived from parameters (activation bounds inside the convolution launch parameters, the
    magnitude_ema gains of eager-mode prep).  Parameters and buffers are fingerprinted at capture (data_ptr, _version); a
    replay after any of them changed raises instead of returning an image of the old weights: re-capture after tuning.
"""
import torch



def _cached_inference_tensors(synthesis):
    """Inference caches derived from the parameters (packed convolution weights, the affine pack, the input's grid / mixing
    matrix, the layers' input gains) were filled during the warm-up and are read by the captured kernels: the graph keeps them
    alive, whatever happens to the caches afterwards."""
    from models.stylegan3 import networks_stylegan3
    from torch_utils.ops import modulated_conv
    return modulated_conv.cached_weight_tensors() + networks_stylegan3.derived_tensors(synthesis)


def _tensors_in(obj, out, seen):
    """Every tensor reachable from `obj` through dicts / lists / tuples / plain objects (PackedConv and friends)."""
    if id(obj) in seen or obj is None or isinstance(obj, (str, bytes, int, float, bool, torch.device, torch.dtype)):
        return out
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors_in(v, out, seen)
    elif isinstance(obj, (list, tuple, set)):
        for v in obj:
            _tensors_in(v, out, seen)
    elif hasattr(obj, '__dict__') and not isinstance(obj, torch.nn.Module):
        _tensors_in(vars(obj), out, seen)
    return out


def _encoder_pack_tensors(encoder):
    """Everything the encoder's captured kernels read besides parameters and buffers: the packed / folded convolution weights of
    the trunk units and heads, the permuted GEMM operands of the head levels, the strip images.  `invalidate_packed()` (train() /
    eval(), .to(), load_state_dict hooks) drops the encoder's references to them; a graph that holds these keeps replaying against
    live memory whose contents still match the unchanged weights (a weight CHANGE is what is_stale() reports)."""
    out, seen = [], set()
    _tensors_in(getattr(encoder, '_packed', None), out, seen)
    for m in encoder.modules():
        if m is not encoder:
            _tensors_in(getattr(m, '_packed', None), out, seen)
    return out


class GraphedSynthesis:
    def __init__(self, generator, batch, all_s_template=None, warmup=2, **synthesis_kwargs):
        """generator: a Generator (or anything with .synthesis / .num_ws / .w_dim) on a CUDA device."""
        self.G = generator
        self.kwargs = dict(noise_mode='const', force_fp32=True)
        self.kwargs.update(synthesis_kwargs)
        dev = next(generator.parameters()).device
        assert dev.type == 'cuda', 'GraphedSynthesis needs the generator on a GPU'
        self.device = dev
        self.use_s = all_s_template is not None
        if self.use_s:
            self.static_s = {k: torch.zeros_like(v, device=dev) for k, v in all_s_template.items()}
            self.static_ws = None
        else:
            self.static_ws = torch.zeros([batch, generator.num_ws, generator.w_dim], device=dev)
        self.batch = int(batch)
        self.static_transform = torch.eye(3, device=dev).repeat(self.batch, 1, 1)
        inp = generator.synthesis.input
        user_transform = inp.transform
        self._load_transform(user_transform)
        inp.transform = self.static_transform                 # shadows the user's tensor during warm-up and capture only
        try:
            # warm up on a side stream (lazy inits, bound caches, allocator), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # thread-local capture: other threads of the process (e.g. the RCCL watchdog of torch.distributed, which polls
            # events) may keep calling into the runtime while this thread captures
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.static_out = self._eager()
        finally:
            inp.transform = user_transform
        self._fingerprint = self._weights_fingerprint()
        self._pinned = _cached_inference_tensors(self.G.synthesis)

    def _load_transform(self, t):
        """Copy `t` ([3,3], [1,3,3] or [batch,3,3]) into the graph's transform buffer (the module keeps the caller's tensor)."""
        if t is self.static_transform:
            return
        t = torch.as_tensor(t, device=self.device, dtype=torch.float32)
        if t.dim() == 3 and t.shape[0] not in (1, self.batch):
            raise ValueError(f'transform batch {t.shape[0]} does not match the captured batch {self.batch}')
        self.static_transform.copy_(t.expand(self.batch, 3, 3) if t.dim() == 3 else t.unsqueeze(0).expand(self.batch, 3, 3))

    def _weights_fingerprint(self):
        inp = self.G.synthesis.input
        return tuple((t.data_ptr(), t._version) for t in list(self.G.synthesis.parameters()) + list(self.G.synthesis.buffers())
                     if t is not self.static_transform and t is not inp.transform)

    def is_stale(self):
        """True when a parameter or buffer of the synthesis network changed (or was replaced) since the capture."""
        return self._weights_fingerprint() != self._fingerprint

    def _eager(self):
        if self.use_s:
            return self.G.synthesis(None, all_s=self.static_s, **self.kwargs)
        return self.G.synthesis(self.static_ws, **self.kwargs)

    def __call__(self, ws=None, all_s=None, transform=None):
        """Copy the inputs (and the current user transform) into the graph's static buffers, replay, return the static output."""
        if self.is_stale():
            raise RuntimeError('GraphedSynthesis: generator parameters / buffers changed since capture (the graph holds their old '
                               'pointers and values derived from them); build a new GraphedSynthesis (is_stale() tells beforehand)')
        self._load_transform(self.G.synthesis.input.transform if transform is None else transform)
        if self.use_s:
            for k, v in all_s.items():
                self.static_s[k].copy_(v, non_blocking=True)
        else:
            self.static_ws.copy_(ws, non_blocking=True)
        self.graph.replay()
        return self.static_out


class GraphedReStyleStep:
    """ONE ReStyle refinement step -- cat(frames, previous output) -> encoder -> + previous latent -> synthesis under the
    identity transform -> face_pool -- captured as a hipGraph for a fixed batch (reference models/setgan/encoder/psp3.py:45-70
    inside utils/inference_utils.py:74-108).  Step 0 is the same graph fed with the average image and `latent_avg`
    (psp3.py:58-60: `codes + latent_avg.repeat(N,1,1)` is the same sum).  An eager step is ~330 launches (150 encoder, 75
    decoder, ~100 small torch ops between them); the replay removes their gaps and all Python between them.

    The encoder's split-precision range guard needs a host read (plain_conv.overflowed), which cannot sit inside a graph: the
    caller resets the flag before the loop and reads it once after it (`run_on_batch` does, and repeats the batch eagerly --
    where the fp32 fallback lives -- if it was raised).  Weight changes after capture raise at the next replay, like
    GraphedSynthesis."""

    def __init__(self, net, batch, warmup=2):
        self.net, self.batch = net, int(batch)
        dev = next(net.parameters()).device
        assert dev.type == 'cuda', 'GraphedReStyleStep needs the network on a GPU'
        self.device = dev
        G = net.decoder
        self.frames = torch.zeros([self.batch, 3, 256, 256], device=dev)
        self.prev_image = torch.zeros([self.batch, 3, 256, 256], device=dev)
        self.prev_latent = torch.zeros([self.batch, int(net.n_styles), int(G.w_dim)], device=dev)
        self.identity = torch.eye(3, device=dev).repeat(self.batch, 1, 1)
        user_transform = G.synthesis.input.transform          # the graph's identity shadows it during warm-up and capture only
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.image, self.latent, self.pooled = self._eager()
        finally:
            G.synthesis.input.transform = user_transform
        self._fingerprint = self._weights_fingerprint()
        self._pinned = _cached_inference_tensors(self.net.decoder.synthesis) + _encoder_pack_tensors(self.net.encoder)

    def _weights_fingerprint(self):
        G = self.net.decoder
        tensors = list(self.net.encoder.parameters()) + list(self.net.encoder.buffers()) + list(G.synthesis.parameters()) + list(G.synthesis.buffers())
        return tuple((t.data_ptr(), t._version) for t in tensors if t is not self.identity and t is not G.synthesis.input.transform)

    def is_stale(self):
        """True when an encoder / generator parameter or buffer changed (or was replaced) since the capture."""
        return self._weights_fingerprint() != self._fingerprint

    def _eager(self):
        net = self.net
        codes = net.encoder._forward_kernels(torch.cat([self.frames, self.prev_image], dim=1)) + self.prev_latent
        net.decoder.synthesis.input.transform = self.identity
        image = net.decoder.synthesis(codes, noise_mode='const', force_fp32=True)
        return image, codes, net.face_pool(image)

    def __call__(self, frames, prev_image, prev_latent, checked=False):
        """Returns (image [B,3,R,R], latent [B,n_styles,512], image pooled to 256^2): static tensors, overwritten by the next replay.
        `checked=True`: the caller has just asked is_stale() itself (run_on_batch does once per batch, not once per step: the
        fingerprint walks ~700 tensors in Python)."""
        if not checked and self.is_stale():
            raise RuntimeError('GraphedReStyleStep: encoder / generator weights changed since capture; build a new one '
                               '(is_stale() tells beforehand; run_on_batch falls back to the eager loop)')
        self.frames.copy_(frames, non_blocking=True)
        self.prev_image.copy_(prev_image, non_blocking=True)
        self.prev_latent.copy_(prev_latent.expand_as(self.prev_latent), non_blocking=True)
        self.graph.replay()                                   # reads the graph's own identity buffer: the module's transform is not touched
        return self.image, self.latent, self.pooled

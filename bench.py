#!/usr/bin/env python3
"""bench.py -- FFHQ-1024 StyleGAN3-T synthesis throughput on MI355X (BASELINE.json configs[1]).

A "step" is one `Generator.synthesis(ws)` forward over a batch of 8 synthetic latents per GPU (seeded random
weights of the config-T 1024 architecture: there are no pretrained weights offline), fp32 execution
(`force_fp32=True`, the mode the 1e-4 parity target is stated for).  Inputs and weights are resident in HBM before the
timed region.  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank runs the same
per-GPU batch on its own images (weak scaling, no data-path collective: images are independent units); the timed
region is bracketed by barrier + synchronize and the MAX over ranks is reported.

One JSON line is printed by rank 0:
  metric/value : whole-job images per second
  roofline     : filtered_lrelu streaming kernel -- algorithmic bytes (C*(in^2+out^2)*4 per image and layer,
                 SURVEY 8d) / its summed launch time measured with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline : the CPU oracle (C/OpenMP port of the reference ref path) timed on this host for ONE image
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch_utils import custom_ops  # noqa: E402

custom_ops.verbosity = 'none'     # stdout carries exactly one line: the JSON result

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0   # same guide: dense fp16 / bf16 MFMA peak (the headline figure with 2:1 sparsity is not used)
ENCODER_GFLOP_PER_IMAGE = 72.3   # IR-SE50 trunk 65.9 + 16 GradualStyleBlock heads 6.4 GFLOP per 256x256 image and ReStyle step


def build_generator(cfg, device):
    from models.stylegan3.networks_stylegan3 import Generator
    from synth_weights import CONFIGS, synth_state_dict
    G = Generator(**CONFIGS[cfg]).eval().requires_grad_(False)
    man = {k: list(v.shape) for k, v in G.state_dict().items()}
    sd = synth_state_dict(man, seed=0, input_bandwidth=float(G.synthesis.input.bandwidth))
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return G.to(device)


def flrelu_algorithmic_bytes(G, batch, elem_size=4):
    """Sum over the synthesis layers that launch filtered_lrelu of C_out * (in^2 + out^2) * sizeof (read the convolution's
    output once, write the activation once).  The ToRGB layer's bias + clamp ride in its convolution in inference: no launch,
    no bytes."""
    total = 0
    per_layer = {}
    for name in G.synthesis.layer_names:
        layer = getattr(G.synthesis, name)
        if layer.fuses_output(torch.float32):
            continue
        k = layer.conv_kernel
        ins = int(layer.in_size[0]) + k - 1
        outs = int(layer.out_size[0])
        nbytes = batch * layer.out_channels * (ins * ins + outs * outs) * elem_size
        per_layer[name] = nbytes
        total += nbytes
    return total, per_layer


def conv_flop_table(G, batch):
    """Per modulated convolution of one synthesis forward (input mix first): (algorithmic FLOP = 2 I O k^2 out^2 per image x batch,
    matrix-core products ISSUED per fp32 product of the algorithm).  Issued factor: 3 for the split-precision kernels (hi*hi + hi*lo +
    lo*hi), 2 where the transform-domain 3x3 kernel runs (12 instead of 18 contractions per output pair, each still three products),
    0 for the ToRGB layer (no matrix cores)."""
    from torch_utils import _sg3abi as abi
    from torch_utils.ops import modulated_conv as mc
    lib = abi.load()
    inp = G.synthesis.input
    rows = [(2.0 * inp.channels * inp.channels * int(inp.size[0]) * int(inp.size[1]) * batch, 3.0)]
    for name in G.synthesis.layer_names:
        layer = getattr(G.synthesis, name)
        k, ci, co, h = layer.conv_kernel, layer.in_channels, layer.out_channels, int(layer.in_size[0])
        s = h + k - 1
        factor = 3.0
        if layer.is_torgb:
            factor = 0.0
        elif k == 3 and mc.f23 != 'off' and mc._f23_wanted(ci, co, h, h, k - 1) and lib.sg3_modconv_f23_supported(abi.SG3_F32, ci, co, h, h, 3, k - 1, 0):
            factor = 2.0
        rows.append((2.0 * ci * co * k * k * s * s * batch, factor))
    return rows


class KernelTimer:
    """Brackets every filtered_lrelu / modulated_conv2d ABI launch with HIP events on the launch stream."""

    def __init__(self):
        self.records = {'filtered_lrelu': [], 'modulated_conv2d': [], 'conv2d_wgrad': []}
        self.enabled = False

    def install(self):
        from torch_utils import _hip_plugins
        from torch_utils.ops import modulated_conv
        timer = self
        orig_f = _hip_plugins.FilteredLreluPlugin.filtered_lrelu
        orig_c = modulated_conv._launch
        orig_w = modulated_conv._weight_gradient

        def timed_wgrad(*a, **k):
            if not timer.enabled:
                return orig_w(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_w(*a, **k)
            e1.record()
            timer.records['conv2d_wgrad'].append((e0, e1))
            return out

        def timed_flrelu(x, *a, **k):
            if not timer.enabled:
                return orig_f(x, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_f(x, *a, **k)
            e1.record()
            timer.records['filtered_lrelu'].append((e0, e1))
            return out

        def timed_conv(x, *a, **k):
            if not timer.enabled:
                return orig_c(x, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_c(x, *a, **k)
            e1.record()
            timer.records['modulated_conv2d'].append((e0, e1))
            return out

        _hip_plugins.FilteredLreluPlugin.filtered_lrelu = staticmethod(timed_flrelu)
        modulated_conv._launch = timed_conv
        modulated_conv._weight_gradient = timed_wgrad

    def reset(self):
        for v in self.records.values():
            v.clear()

    def total_ms(self, key):
        return sum(e0.elapsed_time(e1) for e0, e1 in self.records[key])

    def median_step_ms(self, key, per_step):
        """Median over steps of the summed duration of the `per_step` launches of one step (the first eager step after a
        graph replay pays for fresh allocations inside its brackets; a mean would carry that)."""
        d = [e0.elapsed_time(e1) for e0, e1 in self.records[key]]
        steps = [sum(d[i:i + per_step]) for i in range(0, len(d) - per_step + 1, per_step)]
        return float(np.median(steps)) if steps else 0.0


def cpu_baseline(cfg, runs=3):
    """CPU oracle (oracle/: C + OpenMP restatement of the reference's ref path) on ONE image of the same workload: one warm-up
    forward, then the median of `runs` timed forwards."""
    from helpers import build_oracle_generator
    from oracle import oracle as O
    from synth_weights import synth_ws
    sd, sched = build_oracle_generator(cfg)
    ws = synth_ws(1, sched['num_ws'], sched['w_dim'], seed=1)
    times = []
    for i in range(runs + 1):
        t0 = time.time()
        O.synthesis(sd, sched, ws=ws)
        times.append(time.time() - t0)
    dt = float(np.median(times[1:]))
    return dict(value=1.0 / dt, unit='imgs/s', cores=O.num_threads(), kind='port',
                sample=f'1 image per run, {cfg} synthesis forward fp32; warm-up {times[0]:.1f} s, then median of {runs} runs = {dt:.1f} s '
                       f'({", ".join(f"{t:.1f}" for t in times[1:])}) on {O.num_threads()} OpenMP threads')


def kernel_source_sha():
    """Fingerprint of the filtered_lrelu kernel source: PMC traffic figures are only quoted for the source they were collected on."""
    import hashlib
    with open(os.path.join(ROOT, 'stylegan3-editing_amd', 'csrc', 'sg3_filtered_lrelu.hip'), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def bench_inversion(G, device, rank, world, frames_per_gpu=16, restyle_steps=5, reps=2, decoder='T-1024', timer=None):
    """Secondary measurement (BASELINE metric, second half): ReStyle-pSp inversion frames/s.  Every rank inverts its own
    contiguous range of synthetic 256x256 frames (IR-SE50 encoder with seeded synthetic weights -> 5 refinement steps,
    each one encoder forward + one FFHQ-1024 synthesis forward), then the final latents are all-gathered (RCCL)."""
    import types
    from models.setgan.encoder.psp3 import pSp
    from sg3_runtime.sharded import ShardedInversion
    from synth_weights import synth_encoder_state_dict
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=restyle_steps, resize_outputs=False)
    net = pSp(opts, decoder=G)
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    net.encoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
    net = net.eval().requires_grad_(False).to(device)
    n_frames = frames_per_gpu * world
    frames = torch.from_numpy(np.random.RandomState(9).uniform(-1, 1, size=(n_frames, 3, 256, 256)).astype(np.float32)).to(device)
    inv = ShardedInversion(net, opts, batch_size=frames_per_gpu)
    inv.invert(frames)                                   # warm-up (packs the encoder weights)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        lat, _ = inv.invert(frames)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert tuple(lat.shape) == (n_frames, 16, 512) and bool(torch.isfinite(lat).all())
    # encoder alone (one ReStyle step's encoder forward over this rank's frames): HIP events on the launch stream
    x6 = torch.cat([frames[:frames_per_gpu], frames[:frames_per_gpu]], dim=1)
    with torch.no_grad():
        for _ in range(2):
            net.encoder(x6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            net.encoder(x6)
        e1.record()
        torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / 5
    enc_tflops = ENCODER_GFLOP_PER_IMAGE * frames_per_gpu / enc_ms         # GFLOP / ms = TFLOP/s
    # `achieved` / `frac`: ALGORITHMIC fp32 FLOP rate against the dense fp16 MFMA peak; `issued` / `issued_frac`: the matrix-core work
    # the split-precision kernels issue for it (three fp16 products per fp32 product)
    encoder = {'bound': 'mfma', 'kernel': 'conv2d_f16x3_kernel (IR-SE50 trunk) + head GEMMs', 'achieved': enc_tflops, 'peak': MFMA_F16_PEAK_TFLOPS,
               'unit': 'TFLOP/s', 'frac': enc_tflops / MFMA_F16_PEAK_TFLOPS, 'issued': 3 * enc_tflops, 'issued_frac': 3 * enc_tflops / MFMA_F16_PEAK_TFLOPS,
               'algorithmic_flop_per_image': ENCODER_GFLOP_PER_IMAGE * 1e9, 'ms_per_forward': enc_ms, 'batch': frames_per_gpu}
    # the decoder's filtered_lrelu stages at this batch (config R: the radial instantiations), HIP events around every launch of an
    # eager forward: algorithmic bytes C (in^2 + out^2) 4 per layer / their summed duration
    roofline = None
    if timer is not None:
        ws16 = torch.randn([frames_per_gpu, G.num_ws, G.w_dim], device=device)
        fl_bytes, fl_layers = flrelu_algorithmic_bytes(G, frames_per_gpu)
        with torch.no_grad():
            G.synthesis.input.transform = torch.eye(3, device=device)
            G.synthesis(ws16, noise_mode='const', force_fp32=True)
            timer.reset(); timer.enabled = True
            for _ in range(3):
                G.synthesis(ws16, noise_mode='const', force_fp32=True)
            torch.cuda.synchronize()
            timer.enabled = False
        fl_ms = timer.median_step_ms('filtered_lrelu', len(fl_layers))
        gbs = fl_bytes / (fl_ms * 1e-3) / 1e9 if fl_ms > 0 else 0.0
        radial = any(getattr(G.synthesis, nm).down_radial for nm in fl_layers)
        # PMC traffic of the same launches (tools/sum_traffic.py over `tools/time_config.py R1024 --batch 16`), quoted only for the
        # kernel source and batch it was collected on
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'flrelu_traffic_R.json')
        if radial and frames_per_gpu == 16 and os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get('kernel_source_sha') == kernel_source_sha():
                traffic = tj['traffic_bytes_per_step']
        roofline = {'bound': 'hbm', 'kernel': 'flrelu_stream_kernel' + (' (radial 12x12 down filters)' if radial else ''), 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': traffic, 'algorithmic_bytes_per_forward': fl_bytes,
                    'kernel_ms_per_forward': fl_ms, 'batch': frames_per_gpu}
        timer.reset()
    return dict(metric='ReStyle-pSp video-inversion frames/sec', value=n_frames * reps / float(t.item()), unit='frames/s',
                frames_per_gpu=frames_per_gpu, restyle_steps=restyle_steps, scaling='weak', encoder=encoder, roofline=roofline,
                workload=f'IR-SE50 encoder + FFHQ-1024 config-{decoder} decoder (fp32), 5 ReStyle steps per frame, frames sharded over '
                         'ranks, all-gather of [F,16,512] latents; synthetic weights')


def bench_pti_step(device, cfg='T1024', steps=8, timer=None):
    """One pivotal-tuning step (reference run_pti_images.py:126-139): fp32 synthesis forward with sign write, MSE, backward
    through the fused adjoint / gradient kernels, Adam over the synthesis weights; batch 1, FFHQ-1024."""
    from synth_weights import synth_ws
    G = build_generator(cfg, device)
    G.requires_grad_(True)
    from inversion.scripts.run_pti_images import tuning_optimizer
    opt = tuning_optimizer(list(G.synthesis.parameters())[3:], lr=3e-4)      # Adam as PTI.get_optimizer builds it: one fused launch on a GPU
    w = torch.from_numpy(synth_ws(1, G.num_ws, G.w_dim, seed=3)).to(device)
    target = torch.zeros(1, 3, G.img_resolution, G.img_resolution, device=device)

    def step():
        out = G.synthesis(w, noise_mode='const', force_fp32=True)
        loss = torch.nn.functional.mse_loss(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
    for _ in range(3):                      # the first steps also size the allocator's pools and Adam's state
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {'steps_per_s': 1.0 / dt, 'ms_per_step': dt * 1e3, 'batch': 1}
    if timer is not None:
        # matrix-core work of one step: forward, data gradient (the forward kernels with the scale vectors exchanged) and weight gradient
        # each cost the forward's FLOP (the first layers' data gradients are not needed: counted anyway, a few %); time = the launches
        # of modulated_conv._launch (forward + data gradient) and _weight_gradient between HIP events
        timer.reset(); timer.enabled = True
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        timer.enabled = False
        conv_ms = (timer.total_ms('modulated_conv2d') + timer.total_ms('conv2d_wgrad')) / 2
        rows = conv_flop_table(G, 1)
        alg = 3.0 * sum(f for f, _ in rows)
        issued = sum(f * (2 * k + 3.0 * (k > 0)) for f, k in rows)          # forward + data gradient at the layer's factor, weight gradient at 3
        tf = alg / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out['roofline'] = {'bound': 'mfma', 'kernel': 'modconv (forward, data gradient) + wgrad_f16x3_kernel', 'achieved': tf, 'peak': MFMA_F16_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': tf / MFMA_F16_PEAK_TFLOPS, 'issued_frac': issued / (conv_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS if conv_ms > 0 else 0.0,
                           'algorithmic_flop_per_step': alg, 'kernel_ms_per_step': conv_ms,
                           'wgrad_ms_per_step': timer.total_ms('conv2d_wgrad') / 2}
        timer.reset()
    del G, opt
    torch.cuda.empty_cache()
    return out


def bench_extras(G, ws, device, steps=5, timer=None):
    """Secondary single-GPU measurements (eager launches, not the headline): the reference's default mixed-precision
    execution of the same workload, and config R (1x1 convolutions, radial down filters) at 1024 and 512."""
    from synth_weights import synth_ws

    def run(gen, w, **kw):
        with torch.no_grad():
            for _ in range(2):
                gen.synthesis(w, noise_mode='const', **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gen.synthesis(w, noise_mode='const', **kw)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {'imgs_per_s': int(w.shape[0]) / dt, 'ms_per_step': dt * 1e3, 'batch': int(w.shape[0])}

    G.synthesis.input.transform = torch.eye(3, device=device)     # the inversion measurement leaves per-frame transforms behind
    out = {'T1024_mixed_fp16': run(G, ws)}
    out['T1024_pti_step'] = bench_pti_step(device, 'T1024', timer=timer)
    out['R1024_pti_step'] = bench_pti_step(device, 'R1024', timer=timer)
    for cfg, batch in (('R1024', 4), ('R512', 8)):
        gen = build_generator(cfg, device)
        w = torch.from_numpy(synth_ws(batch, gen.num_ws, gen.w_dim, seed=1)).to(device)
        out[cfg + '_fp32'] = run(gen, w, force_fp32=True)
        if cfg == 'R512':
            # BASELINE configs[4]: StyleCLIP global-direction sweep, 5 betas x 11 alphas of one latent, rendered as
            # StyleSpace batches of 32 (the reference renders the 55 edits one by one)
            from editing.styleclip_global_directions.edit import render_sweep
            with torch.no_grad():
                base = gen.synthesis.W2S(w[:1])
                sweep = {c: v.repeat(55, 1) * (1 + 0.01 * torch.arange(55, device=device).view(-1, 1)) for c, v in base.items()}
                render_sweep(gen, sweep, max_batch=32, force_fp32=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                render_sweep(gen, sweep, max_batch=32, force_fp32=True)
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out['R512_styleclip_sweep'] = {'edits_per_s': 55 / dt, 'ms_per_sweep': dt * 1e3, 'edits': 55, 'max_batch': 32}
        del gen
        torch.cuda.empty_cache()
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start the N ranks as FRESH child processes
    (torch.distributed.run, one rank per GPU, RCCL), relay rank 0's single JSON line and the children's exit status.  This
    parent has not touched the GPU (no HIP call, no torch.cuda.is_available()) and never execs: it only waits."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)          # stderr passes through
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f'[bench] {args.gpus}-rank launch failed: exit code {proc.returncode}, {len(lines)} result lines', file=sys.stderr, flush=True)
        return proc.returncode or 1
    print(lines[0], flush=True)
    return 0


def dry_run(args, rank, world):
    """CPU rehearsal of the multi-rank launch path (gloo, no GPU, a stub in place of the synthesis step): exercises the
    launcher, the rendezvous, the barrier / MAX-over-ranks timing and the one-line output.  Not a measurement: `value` is null."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    x = torch.full([4], float(rank))
    for _ in range(args.warmup):
        x = x * 1.0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = x + 1.0
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    seen = torch.tensor([float(rank)])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({'metric': 'DRY RUN (launch path rehearsal on CPU, gloo) -- not a measurement', 'value': None, 'unit': 'imgs/s',
                          'n_gpus': world, 'world_size_observed': dist.get_world_size() if world > 1 else 1, 'rank_sum': float(seen.item()),
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': float(t.item()) / max(args.steps, 1) * 1e3, 'dry_run': True}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    ap.add_argument('--config', default='T1024')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inversion', action='store_true', help='skip the secondary ReStyle inversion measurement')
    ap.add_argument('--eager', action='store_true', help='launch kernel by kernel instead of replaying a captured hipGraph')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary mixed-precision / config-R measurements')
    ap.add_argument('--dry-run', action='store_true', help='CPU rehearsal of the launch path (gloo, stub step); prints a line with value null')
    args = ap.parse_args(argv)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the driver's scaling run calls `python bench.py --gpus N` directly: this process becomes the launcher (before any GPU use)
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; run `python bench.py --gpus {args.gpus} ...` (it starts the ranks itself) or '
                 f'`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`')
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product has no CPU fallback for the timed path)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)      # 'nccl' IS RCCL on ROCm

    from torch_utils import _sg3abi
    _sg3abi.load()
    from synth_weights import synth_ws
    G = build_generator(args.config, device)
    # each rank owns its own shard of images (independent units; seeds differ per rank)
    ws = torch.from_numpy(synth_ws(args.batch, G.num_ws, G.w_dim, seed=1 + rank)).to(device)

    timer = KernelTimer()
    timer.install()

    def step():
        with torch.no_grad():
            return G.synthesis(ws, noise_mode='const', force_fp32=True)

    launches0 = _sg3abi.launch_count
    graphed = None
    if not args.eager:
        # the timed job replays ONE captured hipGraph per step; per-kernel HIP-event timing for the roofline line is
        # taken from an eager pass after the timed region (events cannot be recorded inside a graph replay)
        from sg3_runtime import GraphedSynthesis
        eager_step = step
        try:
            graphed = GraphedSynthesis(G, args.batch)
        except RuntimeError as err:                     # capture refused (e.g. by a runtime/driver combination): measure eagerly
            print(f'[bench] hipGraph capture failed on rank {rank}, timing eager launches instead: {err}', file=sys.stderr, flush=True)
            graphed = None
            args.eager = True
            torch.cuda.synchronize()
        if graphed is not None:
            def step():  # noqa: F811
                return graphed(ws)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # eager: 30+ ABI launches per forward; graph: the launches were recorded at capture (warm-up + capture passes)
    assert _sg3abi.launch_count - launches0 >= 30, 'HIP kernels did not run'

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = args.eager
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    ksteps = args.steps
    if graphed is not None:
        # same kernels, same inputs, launched eagerly so each launch can be bracketed by events on its stream
        ksteps = max(3, min(args.steps, 10))
        timer.enabled = True
        for _ in range(ksteps):
            eager_step()
        torch.cuda.synchronize()
        timer.enabled = False
    assert tuple(img.shape) == (args.batch, 3, G.img_resolution, G.img_resolution) and bool(torch.isfinite(img).all())
    # per step: the 14 streaming filtered_lrelu launches; 16 convolution launches (the channel mix of the Fourier-feature input runs on
    # the 1x1 kernel too).  Read here: the inversion / PTI legs below reuse (and reset) the timer for their own roofline objects
    total_bytes, fl_layers = flrelu_algorithmic_bytes(G, args.batch)
    fl_ms = timer.median_step_ms('filtered_lrelu', len(fl_layers))
    conv_ms = timer.median_step_ms('modulated_conv2d', len(G.synthesis.layer_names) + 1)
    timer.reset()

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    inversion = inversion_t = None
    if not args.no_inversion and args.config == 'T1024':
        # the reference's default decoder is config R (models/stylegan3/model.py:42-54); the config-T
        # decoder of the headline workload is measured beside it
        G_r = build_generator('R1024', device)
        inversion = bench_inversion(G_r, device, rank, world, decoder='R-1024', timer=timer if rank == 0 else None)
        del G_r
        torch.cuda.empty_cache()
        inversion_t = bench_inversion(G, device, rank, world, decoder='T-1024', timer=None)
    extras = None
    if not args.no_extras and world == 1 and args.config == 'T1024':
        extras = bench_extras(G, ws, device, timer=timer)

    if rank == 0:
        achieved = total_bytes / (fl_ms * 1e-3) / 1e9 if fl_ms > 0 else 0.0
        # HBM traffic of the same 14 launches from PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes over this
        # command, summarised by tools/sum_traffic.py, which stamps the kernel source it ran on); only valid for the default workload
        # and for THIS kernel source -- a stale file reads as null
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'flrelu_traffic.json')
        if os.path.exists(tfile) and args.batch == 8 and args.config == 'T1024':
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get('kernel_source_sha') == kernel_source_sha():
                traffic = tj['traffic_bytes_per_step']
        rows = conv_flop_table(G, args.batch)
        conv_flop = sum(f for f, _ in rows)
        conv_issued = sum(f * k for f, k in rows)
        conv_tflops = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        issued_tflops = conv_issued / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            'metric': 'FFHQ-1024 StyleGAN3-T synthesis imgs/sec', 'value': args.batch * world * args.steps / dt, 'unit': 'imgs/s',
            'n_gpus': world, 'world_size_observed': dist.get_world_size() if world > 1 else 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'StyleGAN3-T FFHQ-1024 Generator.synthesis forward, batch {args.batch} per GPU, force_fp32 '
                                   f'(BASELINE configs[1]); seeded random weights', 'per_gpu_batch': args.batch, 'sharding': 'images',
                       'launch': 'eager' if args.eager else 'hipGraph replay'},
            'roofline': {'bound': 'hbm', 'kernel': 'flrelu_stream_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_step': total_bytes, 'kernel_ms_per_step': fl_ms},
            # second roofline, same shape: the modulated convolutions against the dense fp16 MFMA peak.  `achieved` / `frac` are the
            # ALGORITHMIC fp32 FLOP rate (2 I O k^2 out^2 per image and layer); `issued` / `issued_frac` count the fp16 matrix-core work
            # issued for it: three products per fp32 product in the split-precision kernels, two where the transform-domain 3x3 kernel
            # (Winograd F(2,3) along x: two thirds of the contractions) runs
            'modconv': {'bound': 'mfma', 'kernel': 'modconv_f23_kernel / modconv_f16x3_kernel (+input mix, ToRGB 1x1)', 'achieved': conv_tflops,
                        'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': conv_tflops / MFMA_F16_PEAK_TFLOPS,
                        'issued': issued_tflops, 'issued_frac': issued_tflops / MFMA_F16_PEAK_TFLOPS,
                        'algorithmic_flop_per_step': conv_flop, 'kernel_ms_per_step': conv_ms,
                        'f23_layers': sum(1 for _, k in rows if k == 2.0)},
        }
        out['inversion'] = inversion
        out['inversion_T1024'] = inversion_t
        out['extras'] = extras
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.config)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

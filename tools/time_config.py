"""Time Generator.synthesis for a named config (eager), e.g.  python tools/time_config.py R1024 --batch 2"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch, warnings  # noqa: E402
from helpers import build_product_generator  # noqa: E402
from synth_weights import synth_ws  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument('cfg'); ap.add_argument('--batch', type=int, default=2); ap.add_argument('--iters', type=int, default=3); ap.add_argument('--mixed', action='store_true', help="the reference's default mixed fp16 execution instead of force_fp32")
a = ap.parse_args()
warnings.simplefilter('ignore')
G = build_product_generator(a.cfg, device='cuda:0')
ws = torch.from_numpy(synth_ws(a.batch, G.num_ws, G.w_dim, 1)).cuda()
with torch.no_grad():
    G.synthesis(ws, noise_mode='const', force_fp32=not a.mixed); torch.cuda.synchronize()
    t = time.time()
    for _ in range(a.iters):
        img = G.synthesis(ws, noise_mode='const', force_fp32=not a.mixed)
    torch.cuda.synchronize()
dt = (time.time() - t) / a.iters
print(f'{a.cfg} {"mixed-fp16" if a.mixed else "fp32"} batch {a.batch}: {dt * 1e3:.1f} ms/step  {a.batch / dt:.1f} img/s  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')

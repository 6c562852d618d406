"""Which torch ops a PTI step launches (counts and device time per aten op, per phase):  python tools/profile_pti_ops.py T1024"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch, warnings  # noqa: E402
from torch.profiler import ProfilerActivity, profile, record_function  # noqa: E402
from helpers import build_product_generator  # noqa: E402
from synth_weights import synth_ws  # noqa: E402
warnings.simplefilter('ignore')
cfg = sys.argv[1] if len(sys.argv) > 1 else 'T1024'
G = build_product_generator(cfg, device='cuda:0'); G.requires_grad_(True)
opt = torch.optim.Adam(list(G.synthesis.parameters())[3:], lr=3e-4)
ws = torch.from_numpy(synth_ws(1, G.num_ws, G.w_dim, 1)).cuda()
target = torch.zeros(1, 3, G.img_resolution, G.img_resolution, device='cuda')


def step():
    with record_function('PHASE_forward'):
        out = G.synthesis(ws, noise_mode='const', force_fp32=True)
        loss = torch.nn.functional.mse_loss(out, target)
    with record_function('PHASE_backward'):
        opt.zero_grad(); loss.backward()
    with record_function('PHASE_optimizer'):
        opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
print(f'{"op":60s} {"calls/step":>10s} {"device us/step":>14s} {"cpu us/step":>12s}')
for e in rows[:45]:
    dev = getattr(e, 'device_time_total', None)
    if dev is None:
        dev = getattr(e, 'cuda_time_total', 0.0)
    print(f'{e.key[:60]:60s} {e.count / 3:10.1f} {dev / 3:14.1f} {e.cpu_time_total / 3:12.1f}')

"""Time the ReStyle encoder forward alone:  python tools/time_encoder.py --batch 16"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa: E402
from models.setgan.encoder.encoders.restyle_psp_encoders import BackboneEncoder  # noqa: E402
from synth_weights import synth_encoder_state_dict  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=16); ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
enc = BackboneEncoder(50, 'ir_se', 16)
man = {k: list(v.shape) for k, v in enc.state_dict().items()}
enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
enc = enc.eval().requires_grad_(False).cuda()
x = torch.from_numpy(np.random.RandomState(9).uniform(-1, 1, size=(a.batch, 6, 256, 256)).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(2):
        y = enc(x)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(a.iters):
        y = enc(x)
    torch.cuda.synchronize()
dt = (time.time() - t) / a.iters
print(f'encoder batch {a.batch}: {dt * 1e3:.2f} ms/forward  {a.batch / dt:.1f} frames/s  {72.3 * a.batch / dt / 1e3:.1f} TFLOP/s')

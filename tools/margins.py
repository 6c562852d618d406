"""Print the actual error levels behind a few GPU test tolerances (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch, warnings  # noqa: E402
warnings.simplefilter('ignore')
from helpers import build_product_generator, golden, maxabs  # noqa: E402
from synth_weights import synth_ws  # noqa: E402
DEV = 'cuda:0'
for cfg, gname in (('T1024', 'net_t1024_stats'), ('R512', 'net_r_stats'), ('R1024', 'net_r_stats')):
    g = golden(gname)
    G = build_product_generator(cfg, device=DEV)
    ws = torch.from_numpy(synth_ws(1, G.num_ws, G.w_dim, seed=1)).to(DEV)
    with torch.no_grad():
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
    print(cfg, 'image subsample max-abs error vs reference:', maxabs(img[:1, :, ::16, ::16].cpu().numpy(), g[f'{cfg}/img_sub']), '(tolerance 1e-4)')
    del G
from test_pti_cpu import pti_target, tunable_generator  # noqa: E402
from inversion.scripts.run_pti_images import PTI, default_opts  # noqa: E402
g = golden('pti')
for cfg in ('Ttiny', 'Rtiny'):
    G = tunable_generator(cfg, device=DEV)
    pti = PTI(default_opts(device=DEV, steps=4, learning_rate=3e-3, lpips_lambda=0.0))
    pti.optimize_model(G, synth_ws(1, G.num_ws, G.w_dim, seed=6)[0], pti_target(G.img_resolution, 40))
    print(cfg, 'PTI loss curve max error:', np.abs(np.asarray([h[1] for h in pti.history]) - g[f'{cfg}/image/losses']).max(), '(tolerance 1e-5)')

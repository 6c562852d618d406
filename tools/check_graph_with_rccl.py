"""One-off check: hipGraph capture of the synthesis forward in a process that has an initialised RCCL process group
(its watchdog thread polls events while this thread captures)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch, torch.distributed as dist  # noqa: E402
from helpers import build_product_generator  # noqa: E402
from synth_weights import synth_ws  # noqa: E402
from sg3_runtime import GraphedSynthesis  # noqa: E402
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t = torch.ones(4, device='cuda'); dist.all_reduce(t); dist.barrier()
G = build_product_generator('T256', device='cuda:0')
ws = torch.from_numpy(synth_ws(2, G.num_ws, G.w_dim, 1)).cuda()
g = GraphedSynthesis(G, 2)
a = g(ws).clone()
with torch.no_grad():
    b = G.synthesis(ws, noise_mode='const', force_fp32=True)
dist.barrier()
print('graph replay under an RCCL process group: max diff vs eager', float((a - b).abs().max()))
dist.destroy_process_group()

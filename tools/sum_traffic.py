"""Sum HBM traffic of the filtered_lrelu launches of ONE synthesis forward from two rocprofv3 --pmc passes.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o f -- python bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D -o w -- python bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    python tools/sum_traffic.py D/f_counter_collection.csv D/w_counter_collection.csv profiles/rNN_flrelu_traffic.json
  config R (the inversion decoder, 16 frames per forward -> bench.py's inversion.roofline.traffic):
    ... -- python3 tools/time_config.py R1024 --batch 16 --iters 2   (one pass per counter)
    python tools/sum_traffic.py D/f_... D/w_... profiles/flrelu_traffic_R.json 'one batch-16 R1024 forward'

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a coalesced streaming read, so it is doubled (checked on this kernel: 2 x FETCH_SIZE = 1.09 x the algorithmic read
bytes, the expected halo / warm-up over-fetch; WRITE_SIZE = 1.01 x the algorithmic write bytes).
"""
import csv
import hashlib
import json
import os
import sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _last_forward import flrelu_groups  # noqa: E402


def per_forward(path, counter):
    """Bytes of every filtered_lrelu launch of the LAST forward (14 layers; config R: 16 launches, two layers take a second launch for
    their remainder strips); the ToRGB layer's bias + clamp ride in its convolution."""
    rows = list(csv.DictReader(open(path)))
    names = {int(r['Dispatch_Id']): r['Kernel_Name'] for r in rows}
    val = {int(r['Dispatch_Id']): float(r['Counter_Value']) for r in rows if r['Counter_Name'] == counter}
    groups = flrelu_groups(names)
    per_layer = {lab: sum(val[i] for i in ids) * 1024.0 for lab, ids in groups}
    return sum(per_layer.values()), per_layer, sum(len(ids) for _, ids in groups)


fetch, fetch_l, n_launch = per_forward(sys.argv[1], 'FETCH_SIZE')
fetch *= 2.0
write, write_l, _ = per_forward(sys.argv[2], 'WRITE_SIZE')
what = sys.argv[4] if len(sys.argv) > 4 else 'one batch-8 T1024 forward'
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stylegan3-editing_amd', 'csrc', 'sg3_filtered_lrelu.hip')
sha = hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16]       # bench.py quotes the figure only for this kernel source
out = dict(fetch_bytes_per_step=fetch, write_bytes_per_step=write, traffic_bytes_per_step=fetch + write, kernel_source_sha=sha,
           launches=n_launch, per_layer_bytes={k: 2.0 * fetch_l[k] + write_l[k] for k in fetch_l},
           note=f'{what}, {n_launch} filtered_lrelu launches; FETCH_SIZE doubled per the gfx950 correction')
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)

"""Sum HBM traffic of the filtered_lrelu launches of ONE synthesis forward from two rocprofv3 --pmc passes.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o f -- python bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D -o w -- python bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    python tools/sum_traffic.py D/f_counter_collection.csv D/w_counter_collection.csv profiles/rNN_flrelu_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a coalesced streaming read, so it is doubled (checked on this kernel: 2 x FETCH_SIZE = 1.09 x the algorithmic read
bytes, the expected halo / warm-up over-fetch; WRITE_SIZE = 1.01 x the algorithmic write bytes).
"""
import csv
import hashlib
import json
import os
import sys


def per_forward(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'flrelu_' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    last = rows[-14:]            # the 14 streaming launches of the last forward (ToRGB's bias + clamp ride in its convolution)
    assert len(last) == 14 and all('stream' in r['Kernel_Name'] for r in last), len(last)
    return sum(float(r['Counter_Value']) for r in last) * 1024.0


fetch = 2.0 * per_forward(sys.argv[1], 'FETCH_SIZE')
write = per_forward(sys.argv[2], 'WRITE_SIZE')
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stylegan3-editing_amd', 'csrc', 'sg3_filtered_lrelu.hip')
sha = hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16]       # bench.py quotes the figure only for this kernel source
out = dict(fetch_bytes_per_step=fetch, write_bytes_per_step=write, traffic_bytes_per_step=fetch + write, kernel_source_sha=sha,
           note='one batch-8 T1024 forward, 14 filtered_lrelu launches; FETCH_SIZE doubled per the gfx950 correction')
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)

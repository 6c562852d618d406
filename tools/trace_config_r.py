"""Per-layer summary of a rocprofv3 kernel trace of `tools/time_config.py R1024 --batch B` (last forward pass):
   python tools/trace_config_r.py <kernel_trace.csv> [batch]"""
import csv
import sys

# (in_channels, out_channels, in_size, out_size) of config R-1024 (synthesis_schedule(1024, 3, 65536, 1024))
LAYERS = [(1024, 1024, 36, 36), (1024, 1024, 36, 36), (1024, 1024, 36, 52), (1024, 1024, 52, 52), (1024, 1024, 52, 84),
          (1024, 1024, 84, 148), (1024, 1024, 148, 148), (1024, 645, 148, 276), (645, 406, 276, 276), (406, 256, 276, 532),
          (256, 161, 532, 1044), (161, 102, 1044, 1044), (102, 64, 1044, 1044), (64, 64, 1044, 1024), (64, 3, 1024, 1024)]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sel = [r for r in rows if 'sg3::' in r['Kernel_Name'] and 'prep' not in r['Kernel_Name'] and 'fourier' not in r['Kernel_Name']]
# The last forward: 15 convolutions (+ the input's channel mix) back from the end.  A layer's filtered_lrelu may be two launches
# (12x12 down filter at up 2 on 148- / 276-column rows: full strips, then the packed remainder strip); the ToRGB layer's rides
# in its convolution in inference.
groups = []                                   # [is_conv, name, duration us]
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    conv = 'modconv' in r['Kernel_Name']
    if groups and not conv and not groups[-1][0]:
        groups[-1][2] += d; groups[-1][1] += ' + remainder launch'
    else:
        groups.append([conv, r['Kernel_Name'][10:52], d])
n_conv, start = 0, len(groups)
while start > 0 and n_conv < 15:
    start -= 1
    n_conv += groups[start][0]
ci = fi = 0
tc = tf = 0.0
for conv, n, d in groups[start:]:
    if conv:
        i, o, s, _ = LAYERS[ci]
        fl = 2.0 * i * o * s * s * B
        by = 4.0 * (i + o) * s * s * B
        print(f"conv L{ci:<2d} {i:4d}->{o:4d} @{s:4d}  {n:42s} {d:8.1f} us {fl / d / 1e6:7.1f} TF/s {by / d / 1e6:6.2f} TB/s")
        ci += 1; tc += d
    else:
        i, o, s, so = LAYERS[fi]
        by = 4.0 * o * (s * s + so * so) * B
        print(f"flr  L{fi:<2d} {o:4d} ch {s:4d}->{so:4d}  {n:42s} {d:8.1f} us {by / d / 1e6:6.2f} TB/s")
        fi += 1; tf += d
print(f'conv total {tc / 1e3:.2f} ms   flrelu total {tf / 1e3:.2f} ms')

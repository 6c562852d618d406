"""Summarise a rocprofv3 kernel trace of bench.py per synthesis layer (last forward pass)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last forward: every convolution / filtered_lrelu launch after the last Fourier-feature kernel, minus the input's channel mix
# (the first convolution).  The ToRGB layer's filtered_lrelu rides in its convolution in inference (a call with hooks on the
# layers keeps it a launch of its own: flrelu_pointwise_kernel).
start = max(i for i, r in enumerate(rows) if 'fourier_features' in r['Kernel_Name'])
last = [r for r in rows[start:] if 'sg3' in r['Kernel_Name'] and      # fp16 instantiations come out mangled (_ZN3sg3...)
         ('modconv' in r['Kernel_Name'] or 'flrelu' in r['Kernel_Name'])
        and 'prep' not in r['Kernel_Name']][1:]
gf = [6.81, 6.81, 6.81, 13.76, 13.76, 34.90, 106.17, 66.98, 91.21, 36.15, 53.22, 81.36, 32.14, 20.17, 0.20]
mb = [5.6, 5.6, 8.5, 11.5, 20.4, 60.0, 90.9, 127.5, 124.6, 184.5, 445.5, 445.5, 279.6, 274.3, 25.2]      # MB per image at fp32 I/O
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ci = fi = 0
tc = tf = 0.0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name']
    if 'modconv' in n:
        print(f"conv L{ci:<2d} {n[10:58]:48s} {d:9.1f} us {gf[ci] * B / d * 1e3:7.1f} TF/s  grid={r['Grid_Size_X']:>9s} lds={r['LDS_Block_Size']}")
        ci += 1; tc += d
    else:
        print(f"flr  L{fi:<2d} {n[10:58]:48s} {d:9.1f} us {mb[fi] * B / d:7.2f} TB/s  grid={r['Grid_Size_X']:>9s}")
        fi += 1; tf += d
print(f'conv total {tc / 1e3:.2f} ms   flrelu total {tf / 1e3:.2f} ms')

"""Per-layer kernel-only time of wgrad_f16x3_kernel and of the partial-sum reduction from a rocprofv3 kernel trace of tools/bench_wgrad.py:
    rocprofv3 --kernel-trace --output-format csv -d D -o w -- python3 tools/bench_wgrad.py T1024 ; python tools/wgrad_trace.py D/.../w_kernel_trace.csv"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
per = 13                                                # 3 warm-up + 10 timed calls per layer
wg = [r for r in rows if 'wgrad_f16x3' in r['Kernel_Name']]
red = [r for r in rows if 'reduce_kernel' in r['Kernel_Name']]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3  # noqa: E731
tk = tr = 0.0
for i in range(0, len(wg) - per + 1, per):
    k = sorted(dur(r) for r in wg[i + 3:i + per])[5]
    g = wg[i]
    grid = f"{g.get('Grid_Size_X', g.get('Grid_Size', '?'))}x{g.get('Grid_Size_Y', '')}x{g.get('Grid_Size_Z', '')}"
    rr = [dur(r) for r in red if int(wg[i + 3]['Start_Timestamp']) < int(r['Start_Timestamp']) < int(wg[i + per - 1]['End_Timestamp']) + 10 ** 6]
    r_ = sorted(rr)[len(rr) // 2] if rr else 0.0
    tk += k; tr += r_
    print(f'layer {i // per:2d}: kernel {k:8.1f} us   reduction {r_:7.1f} us   grid {grid}')
print(f'total kernel {tk:.1f} us, reductions {tr:.1f} us')

"""Per-launch VALU issue statistics of the filtered_lrelu launches of ONE synthesis forward, from a rocprofv3 --pmc pass.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \\
        --output-format csv -d D -o v -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    python tools/sum_valu.py D/v_counter_collection.csv D/v_kernel_trace.csv > profiles/rNN_flrelu_valu_pmc.txt
    ... -- python3 tools/time_config.py R1024 --batch 8 --iters 2   +   sum_valu.py <csv> <trace> "config R-1024"  for the radial kernels

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over
waves; one VALU instruction holds its SIMD's vector issue for 4 cycles = 1 quad-cycle, so SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU here.
GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration.  VALU issue utilisation =
4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _last_forward import flrelu_groups  # noqa: E402

groups = flrelu_groups({k: v['name'] for k, v in by.items()})
title = sys.argv[3] if len(sys.argv) > 3 else 'T-1024'
n_launch = sum(len(d) for _, d in groups)
print(f'filtered_lrelu launches of one {title} forward: {n_launch} launches for 14 layers (profiled pass: clocks read 3-5 % below an un-profiled run)')
print(f"{'layer':22s} {'kernel':22s} {'us':>8s} {'GHz':>5s} {'VALU insts M':>13s} {'VALU issue util':>16s} {'waves/SIMD':>11s} {'wait_any':>9s} {'wait_inst':>10s}")
tv = ta = 0.0
for lab, ids in groups:
    for j, k in enumerate(ids):
        d = by[k]
        us, act = dur[k], d['GRBM_GUI_ACTIVE'] / 8
        iv, av, wc = d['SQ_INSTS_VALU'], d['SQ_ACTIVE_INST_VALU'], d['SQ_WAVE_CYCLES']
        tv += av * 4 / 1024; ta += act
        kern = d['name'][d['name'].index('<'):d['name'].index('>') + 1]
        name = lab if len(ids) == 1 else lab + (' full strips' if j == 0 else ' remainder strips')
        print(f"{name:22s} {kern:22s} {us:8.1f} {act / us / 1e3:5.2f} {iv / 1e6:13.1f} {av * 4 / 1024 / act:16.2f} {wc * 4 / 1024 / act:11.2f} "
              f"{d['SQ_WAIT_ANY'] / wc:9.2f} {d['SQ_WAIT_INST_ANY'] / wc:10.2f}")
print(f'all {n_launch} launches: VALU issue utilisation {tv / ta:.2f} at the clock each launch ran at')

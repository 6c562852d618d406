#!/bin/bash
# diagnostics of the transform-domain kernel: in-kernel segment stamps and per-launch SQ counters (L6, direct vs F(2,3))
set -e
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=$(pwd)/gpurun_out/f23diag; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSG3_F23_STAMPS tools/f23_stamps.hip -o /tmp/f23s 2>/dev/null
for tn in 4 5 7; do echo "== stamps TN=$tn"; /tmp/f23s $tn; done > $OUT/stamps.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SG3_F23_TN=5 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_f23 -o p -- python3 tools/bench_layer.py conv L6 L8 --f23 on --iters 3 > $OUT/pmc_f23.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_dir -o p -- python3 tools/bench_layer.py conv L6 L8 --f23 off --iters 3 > $OUT/pmc_dir.log 2>&1
python3 - <<'PY' > $OUT/pmc_summary.txt
import csv, glob, os
for tag in ('pmc_f23', 'pmc_dir'):
    base = os.path.join(os.environ.get('OUT', 'gpurun_out/f23diag'), tag)
    cc = glob.glob(base + '/**/p_counter_collection.csv', recursive=True)[0]
    kt = glob.glob(base + '/**/p_kernel_trace.csv', recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(cc)):
        by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
    dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
    print(tag)
    for k, d in sorted(by.items()):
        if 'modconv' not in d['name'] or 'prep' in d['name']:
            continue
        act = d['GRBM_GUI_ACTIVE'] / 8; wc = d['SQ_WAVE_CYCLES']
        print(f"  {d['name'][:60]:60s} {dur[k]:8.1f} us  {act / dur[k] / 1e3:4.2f} GHz  mfma util {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * act):4.2f}  valu issue {d['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * act):4.2f} "
              f" valu insts {d['SQ_INSTS_VALU'] / 1e6:7.1f} M  waves/simd {wc * 4 / 1024 / act:4.2f}  wait_any {d['SQ_WAIT_ANY'] / wc:4.2f}  wait_inst {d['SQ_WAIT_INST_ANY'] / wc:4.2f} "
              f" lds active {d['SQ_ACTIVE_INST_LDS'] * 4 / (1024 * act):4.2f}  bank conflict {d['SQ_LDS_BANK_CONFLICT'] / act / 256:5.3f}")
PY
cat $OUT/stamps.txt $OUT/pmc_summary.txt

"""Per-dispatch SQ counters of the kernels whose name contains a pattern, from one rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \\
        SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d D -o p -- python3 <program>
    python tools/pmc_summary.py D <pattern> [every]
Units as in tools/sum_valu.py (quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
import csv
import glob
import sys

base, pat = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cc = glob.glob(base + '/**/*_counter_collection.csv', recursive=True)[0]
kt = glob.glob(base + '/**/*_kernel_trace.csv', recursive=True)[0]
by = {}
for r in csv.DictReader(open(cc)):
    by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
sel = [(k, d) for k, d in sorted(by.items()) if pat in d['name']]
for k, d in sel[::every]:
    act = d['GRBM_GUI_ACTIVE'] / 8
    wc = d['SQ_WAVE_CYCLES']
    print(f"{d['name'][:48]:48s} {dur[k]:8.1f} us  {act / dur[k] / 1e3:4.2f} GHz  mfma busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * act):4.2f}  valu issue {d['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * act):4.2f} "
          f" valu insts {d['SQ_INSTS_VALU'] / 1e6:7.1f} M  waves/simd {wc * 4 / 1024 / act:4.2f}  wait_any {d['SQ_WAIT_ANY'] / wc:4.2f}  wait_inst {d['SQ_WAIT_INST_ANY'] / wc:4.2f} "
          f" lds active {d['SQ_ACTIVE_INST_LDS'] * 4 / (1024 * act):4.2f}  bank conflict {d['SQ_LDS_BANK_CONFLICT'] / act / 256:5.3f}")

"""Summary of tools/f23_pmc.sh: per dispatch of modconv_f23_kernel -- duration, clock, matrix-pipe busy share, share of SIMD cycles in
which matrix and vector instructions executed together (SQ_VALU_MFMA_COEXEC_CYCLES), vector issue share, waves per SIMD.
Units as in tools/pmc_summary.py (GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_* summed over the chip)."""
import csv
import glob
import sys

base = sys.argv[1]
cc = glob.glob(base + '/**/*_counter_collection.csv', recursive=True)[0]
kt = glob.glob(base + '/**/*_kernel_trace.csv', recursive=True)[0]
by = {}
for r in csv.DictReader(open(cc)):
    by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
for k, d in sorted(by.items()):
    if 'modconv' not in d['name'] or 'prep' in d['name']:
        continue
    act = d['GRBM_GUI_ACTIVE'] / 8
    wc = d['SQ_WAVE_CYCLES']
    print(f"{d['name'][10:40]:30s} {dur[k]:8.1f} us  {act / dur[k] / 1e3:4.2f} GHz  mfma busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * act):5.3f}  "
          f"mfma+valu coexec {d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / (1024 * act):5.3f}  valu issue {d['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * act):5.3f}  "
          f"valu insts {d['SQ_INSTS_VALU'] / 1e6:7.1f} M  waves/simd {wc * 4 / 1024 / act:4.2f}  wait_any {d['SQ_WAIT_ANY'] / wc:4.2f}")

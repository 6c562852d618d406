"""Matrix-core utilisation of the modulated-convolution launches of ONE synthesis forward, from a rocprofv3 --pmc pass.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d D -o m -- \\
        python bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras
    python tools/sum_mfma.py D/m_counter_collection.csv > profiles/rNN_conv_mfma_util.txt

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip, the cycles a SIMD's matrix pipe is busy (32 per
v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE the cycles of the dispatch at the clock the chip ran at.
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (L6: 32.96 M for a 2.3 ms dispatch = 8 x 4.12 M cycles at 1.78 GHz), so
utilisation = busy / (1024 SIMDs x active / 8)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
disp = [v for k, v in sorted(by.items()) if 'modconv' in v['name'] and 'prep' not in v['name']]
last = disp[-16:]                # input mix + 15 layers of the last forward
tb = ta = 0.0
for j, d in enumerate(last):
    busy, act = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), d.get('GRBM_GUI_ACTIVE', 0.0)
    tb += busy; ta += act
    label = 'input mix' if j == 0 else f'L{j - 1}'
    print(f"{label:10s} {d['name'][10:66]:56s} MFMA-busy {busy / 1e6:9.2f} Mcycles  active {act / 8e3:8.1f} kcycles per XCD  utilisation {busy / (1024 * act / 8) if act else 0:5.2f}")
print(f"all convolutions of one forward: utilisation {tb / (1024 * ta / 8):.2f} of the matrix pipes at the clock the chip sustained")

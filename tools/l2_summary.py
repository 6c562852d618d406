"""L2 statistics per dispatch of the kernels whose name contains a pattern, from two rocprofv3 --pmc passes of the same program
(TCC counters do not all fit one pass):  python tools/l2_summary.py DIR_HITMISS DIR_FETCH <pattern>
    pass 1: --pmc TCC_HIT_sum TCC_MISS_sum        pass 2: --pmc FETCH_SIZE
FETCH_SIZE is printed as reported (KiB -> bytes); on gfx950 it tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md), so the
bytes that left the L2 are up to twice the figure."""
import csv
import glob
import sys


def load(base):
    cc = glob.glob(base + '/**/*_counter_collection.csv', recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(cc)):
        by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
    return by


a, b, pat = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
ka = [d for k, d in sorted(a.items()) if pat in d['name']]
kb = [d for k, d in sorted(b.items()) if pat in d['name']]
for da, db in zip(ka, kb):
    hit, miss = da.get('TCC_HIT_sum', 0.0), da.get('TCC_MISS_sum', 0.0)
    print(f"{da['name'][:50]:50s} L2 requests {(hit + miss) / 1e6:8.1f} M  hit rate {hit / max(hit + miss, 1):5.3f}  FETCH_SIZE {db.get('FETCH_SIZE', 0.0) * 1024 / 1e9:7.3f} GB")

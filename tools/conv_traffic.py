"""HBM traffic of the convolution launches of ONE synthesis forward from two rocprofv3 --pmc passes, launch by launch, beside the
algorithmic bytes of each (input read once + output written once, fp32):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o f -- python3 tools/time_config.py R1024 --batch 8 --iters 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D -o w -- python3 tools/time_config.py R1024 --batch 8 --iters 2
    python tools/conv_traffic.py D/f_counter_collection.csv D/w_counter_collection.csv R1024 8

Counters in KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section), as in tools/sum_traffic.py."""
import csv
import sys

CHANNELS = {'R1024': ([1024] * 7 + [645, 406, 256, 161, 102, 64, 64, 3], [36, 36, 52, 52, 84, 148, 148, 276, 276, 532, 1044, 1044, 1044, 1024, 1024], 0),
            'T1024': ([512] * 7 + [323, 203, 128, 81, 51, 32, 32, 3], [36, 36, 52, 52, 84, 148, 148, 276, 276, 532, 1044, 1044, 1044, 1024, 1024], 2)}


def load(path, counter):
    rows = list(csv.DictReader(open(path)))
    names = {int(r['Dispatch_Id']): r['Kernel_Name'] for r in rows}
    val = {int(r['Dispatch_Id']): float(r['Counter_Value']) * 1024.0 for r in rows if r['Counter_Name'] == counter}
    return names, val


def main():
    fpath, wpath, cfg, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    names, fetch = load(fpath, 'FETCH_SIZE')
    _, write = load(wpath, 'WRITE_SIZE')
    conv = [i for i in sorted(names) if 'modconv' in names[i] and 'prep' not in names[i] and 'transpose' not in names[i]]
    conv = conv[-15:]                                                    # the last forward: L0 .. L14
    chans, sizes, grow = CHANNELS[cfg]
    cin = [chans[0]] + chans[:-1]
    print(f'{cfg} batch {batch}: convolution launches of the last forward; traffic = 2 x FETCH_SIZE + WRITE_SIZE')
    tot_t = tot_a = 0.0
    for l, i in enumerate(conv):
        hin = sizes[l - 1] if l else 36
        # the convolution of layer l reads the previous layer's output (in_size = sizes[l-1]; 36 for L0) and writes in_size + k - 1
        hout = hin + grow if l < 14 else hin
        alg = batch * 4.0 * (cin[l] * hin * hin + chans[l] * hout * hout)
        t = 2.0 * fetch[i] + write[i]
        tot_t += t; tot_a += alg
        print(f'L{l:<2d} {cin[l]:4d}->{chans[l]:4d} @{hin:4d}  read {2 * fetch[i] / 1e9:7.3f} GB  written {write[i] / 1e9:7.3f} GB  traffic {t / 1e9:7.3f} GB  '
              f'algorithmic {alg / 1e9:7.3f} GB  x{t / alg:5.2f}   {names[i][:48]}')
    print(f'total traffic {tot_t / 1e9:.2f} GB, algorithmic {tot_a / 1e9:.2f} GB, x{tot_t / tot_a:.2f}')


if __name__ == '__main__':
    main()

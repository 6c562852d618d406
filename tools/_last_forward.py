"""Shared by the PMC summarisers: the filtered_lrelu launches of the LAST synthesis forward in a rocprofv3 counter / trace CSV,
grouped by layer.  A forward is the input's channel mix + 15 layer convolutions (16 `modconv*` launches that are not prep kernels);
every layer but ToRGB is followed by its filtered_lrelu -- ONE launch, or TWO where the 12x12 down filter at up 2 meets 148- / 276-
column rows (config R: full strips, then the packed remainder strips).  Returns [(label, [dispatch ids])], labels 'L0'..'L13'."""


def is_conv(name):
    return 'modconv' in name and 'prep' not in name


def flrelu_groups(names_by_dispatch):
    """names_by_dispatch: {dispatch id: kernel name} of the whole run."""
    ids = sorted(names_by_dispatch)
    convs = [i for i in ids if is_conv(names_by_dispatch[i])]
    assert len(convs) >= 16, f'{len(convs)} convolution launches: not a whole forward'
    first = convs[-16]
    groups, layer, fresh = [], -1, False
    for i in ids:
        if i < first:
            continue
        nm = names_by_dispatch[i]
        if is_conv(nm):
            layer += 1                      # -1 -> 0 is the input mix; layer k's convolution makes layer = k + 1
            fresh = True
        elif 'flrelu_stream' in nm:
            if fresh:
                groups.append([f'L{layer - 1}', [i]])
            else:
                groups[-1][1].append(i)     # second launch of the same layer: the remainder strips
            fresh = False
    assert len(groups) == 14, [g[0] for g in groups]
    return [(lab, d) for lab, d in groups]

"""Host-side helpers of the profiling tools that decide which launches a judged number is summed over (`tools/_last_forward.py`:
used by tools/sum_valu.py and tools/sum_traffic.py for `roofline.traffic` / `inversion.roofline.traffic`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _forward(names, start, remainder_layers=()):
    i = start
    names[i] = 'void sg3::modconv1_f16x3_kernel<float, 2, 4, 4, 2, true, 2, true>(sg3::ConvParams)'; i += 1        # the input's channel mix
    for layer in range(15):
        names[i] = 'sg3::modconv_prep_s_batch_kernel(sg3::PrepBatch)'; i += 1                                          # prep kernels do not count
        names[i] = ('void sg3::modconv_f23_kernel<7, float>(sg3::F23Params)' if layer < 14 else
                    'void sg3::modconv_1x1_small_kernel<float, 4>(sg3::ConvParams, int)'); i += 1
        if layer < 14:
            names[i] = 'void sg3::flrelu_stream_kernel<float, 2, 2, 0, 5, 0, 1>(sg3::StreamParams)'; i += 1
            if layer in remainder_layers:
                names[i] = 'void sg3::flrelu_stream_kernel<float, 2, 2, 0, 5, 0, 2>(sg3::StreamParams)'; i += 1       # remainder strips
    return i


def test_last_forward_groups_launches_by_layer():
    from _last_forward import flrelu_groups
    names = {}
    nxt = _forward(names, 1)                              # a warm-up forward (config T: one launch per layer)
    names[nxt] = 'void at::native::vectorized_elementwise_kernel<4>'; nxt += 1
    first_of_last = nxt
    _forward(names, nxt, remainder_layers=(6, 8))         # the last forward (config R: L6 / L8 take a second launch)
    groups = flrelu_groups(names)
    assert [g[0] for g in groups] == [f'L{j}' for j in range(14)]
    assert [len(g[1]) for g in groups] == [2 if j in (6, 8) else 1 for j in range(14)]
    assert all(i >= first_of_last for _, ids in groups for i in ids)          # nothing of the warm-up forward
    assert sum(len(ids) for _, ids in groups) == 16


def test_committed_traffic_files_belong_to_the_current_kernel_source():
    """bench.py quotes `roofline.traffic` / `inversion.roofline.traffic` from profiles/flrelu_traffic*.json only when their stamp matches
    csrc/sg3_filtered_lrelu.hip; a source edit silently turns both fields into null.  This test is the reminder to re-collect
    (gpurun -- 'bash tools/collect_traffic.sh', then copy the two files into profiles/)."""
    import hashlib
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'stylegan3-editing_amd', 'csrc', 'sg3_filtered_lrelu.hip'), 'rb') as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    for name in ('flrelu_traffic.json', 'flrelu_traffic_R.json'):
        with open(os.path.join(root, 'profiles', name)) as f:
            assert json.load(f)['kernel_source_sha'] == sha, f'profiles/{name} was collected on another version of the kernel source'

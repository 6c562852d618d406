#!/bin/bash
# Round-3 left one wrong result unexplained: a buffer_load into ACCUMULATOR registers (the next tile's A fragments) issued between the
# K loop's closing wait and the output transform's v_accvgpr_reads "returned NaN in every output".  This rebuilds the three placements
# (F23_AGPR_PROBE = 1 in front of the closing wait | 2 behind it | 3 behind it with its own vmcnt(0)) with an out-of-range request and
# runs the transform-domain tests on each.  Wrong results only (range-checked loads): no fault risk.
mkdir -p gpurun_out/agpr_probe
for m in 1 2 3; do
  touch stylegan3-editing_amd/csrc/sg3_modconv_f23.hip
  make -C stylegan3-editing_amd/csrc EXTRA=-DF23_AGPR_PROBE=$m > /dev/null 2>&1
  echo "== F23_AGPR_PROBE=$m"
  timeout -k 10 300 python -m pytest tests/test_gpu_f23.py -q -x 2>&1 | tail -4
done
touch stylegan3-editing_amd/csrc/sg3_modconv_f23.hip
make -C stylegan3-editing_amd/csrc > /dev/null 2>&1
echo "== default build"; timeout -k 10 300 python -m pytest tests/test_gpu_f23.py -q 2>&1 | tail -2

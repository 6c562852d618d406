#!/bin/bash
# same-box A/B of the F(2,3) staging arithmetic: single-issue fp32 (default) vs the packed-fp32 form of round 3
set -e
mkdir -p gpurun_out
L="L3 L5 L6 L7 L8 L9"
echo "== single-issue (default build)"; python tools/bench_layer.py conv $L
touch stylegan3-editing_amd/csrc/sg3_modconv_f23.hip
make -C stylegan3-editing_amd/csrc EXTRA=-DF23_STAGE_PACKED=1 > /dev/null 2>&1
echo "== packed fp32 (round 3)"; python tools/bench_layer.py conv $L
touch stylegan3-editing_amd/csrc/sg3_modconv_f23.hip
make -C stylegan3-editing_amd/csrc > /dev/null 2>&1
echo "== single-issue again"; python tools/bench_layer.py conv $L

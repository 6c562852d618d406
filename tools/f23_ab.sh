#!/bin/bash
# A/B of the transform-domain 3x3 kernel on the T-1024 layer shapes (batch 8): direct kernel vs F(2,3) at each rows-per-wave
set -e
mkdir -p gpurun_out
L="L3 L4 L5 L6 L7 L8 L9 L10 L11"
echo "== direct" ; python tools/bench_layer.py conv $L --f23 off
for tn in 4 5 7; do echo "== f23 TN=$tn"; SG3_F23_TN=$tn python tools/bench_layer.py conv $L --f23 on; done

#!/bin/bash
# wave priority variants of the transform-domain kernel: 0 none | 1 MFMA phase at priority 1 | 2 waves 4-7 at priority 1 throughout | 3 staging block at priority 1
# | 4 waves 0-3 at priority 1 throughout   (MODES="4 2" bash tools/f23_policy.sh)
cd stylegan3-editing_amd/csrc
for mode in ${MODES:-1 0 2 3}; do
  touch sg3_modconv_f23.hip; make EXTRA="-DF23_PRIO_MODE=$mode" > /tmp/mk.log 2>&1 || tail -3 /tmp/mk.log
  for tn in 5 7; do
    echo "== priority mode $mode TN=$tn"
    (cd ../.. && SG3_F23_TN=$tn python tools/bench_layer.py conv L5 L6 L8 L9 --f23 on 2>&1 | grep -v amdgpu.ids)
  done
done

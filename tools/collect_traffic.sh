#!/bin/bash
# Re-collects the two PMC traffic files bench.py quotes (profiles/flrelu_traffic.json, profiles/flrelu_traffic_R.json) for the CURRENT
# csrc/sg3_filtered_lrelu.hip (the files carry the source's sha; bench.py reports `traffic: null` for any other source).
#     gpurun --timeout 900 -- 'bash tools/collect_traffic.sh'      -> gpurun_out/traffic/{flrelu_traffic.json,flrelu_traffic_R.json}
# Separate passes per counter, --kernel-trace only, the program itself after `--` (MI355X_MICROARCH.md, HBM section).
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/traffic
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
first() { ls "$@" 2>/dev/null | head -1; }
EAGER="--eager --steps 2 --warmup 1 --no-inversion --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/Tf" -o f -- python3 bench.py $EAGER > /dev/null 2> "$OUT/Tf.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/Tw" -o w -- python3 bench.py $EAGER > /dev/null 2> "$OUT/Tw.err"
python3 tools/sum_traffic.py "$(first $OUT/Tf/*/f_counter_collection.csv $OUT/Tf/f_counter_collection.csv)" \
                             "$(first $OUT/Tw/*/w_counter_collection.csv $OUT/Tw/w_counter_collection.csv)" "$OUT/flrelu_traffic.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/Rf" -o f -- python3 tools/time_config.py R1024 --batch 16 --iters 2 > /dev/null 2> "$OUT/Rf.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/Rw" -o w -- python3 tools/time_config.py R1024 --batch 16 --iters 2 > /dev/null 2> "$OUT/Rw.err"
python3 tools/sum_traffic.py "$(first $OUT/Rf/*/f_counter_collection.csv $OUT/Rf/f_counter_collection.csv)" \
                             "$(first $OUT/Rw/*/w_counter_collection.csv $OUT/Rw/w_counter_collection.csv)" "$OUT/flrelu_traffic_R.json" "one batch-16 R1024 forward"
rm -rf "$OUT/Tf" "$OUT/Tw" "$OUT/Rf" "$OUT/Rw"
ls -la "$OUT"

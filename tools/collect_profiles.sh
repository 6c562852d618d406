#!/bin/bash
# Collects the judged profile set of a round on the GPU box (one MI355X).  Run from the repo root through gpurun:
#     gpurun --timeout 1150 -- 'bash tools/collect_profiles.sh r02'
# Writes gpurun_out/prof_<tag>/... ; copy what is listed at the end into profiles/ (see profiles/README.md).
# Every profiler run has the program itself after `--` (no env / bash -c hop), PMC passes carry --kernel-trace only.
set -eo pipefail
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
SHORT="--no-inversion --no-extras --no-cpu-baseline"
EAGER="--eager --steps 2 --warmup 1 $SHORT"

echo "[1/6] kernel trace + stats of the default bench (short form)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o b -- python3 bench.py $SHORT > "$OUT/bench_line_short.json" 2> "$OUT/trace.err"
python3 tools/trace_layers.py "$(ls $OUT/trace/*/b_kernel_trace.csv $OUT/trace/b_kernel_trace.csv 2>/dev/null | head -1)" > "$OUT/bench_per_layer.txt"
cp "$(ls $OUT/trace/*/b_kernel_stats.csv $OUT/trace/b_kernel_stats.csv 2>/dev/null | head -1)" "$OUT/bench_kernel_stats.csv"
rm -rf "$OUT/trace"

echo "[2/6] HBM traffic of the filtered_lrelu launches (two PMC passes)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc" -o f -- python3 bench.py $EAGER > /dev/null 2> "$OUT/pmc_f.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc" -o w -- python3 bench.py $EAGER > /dev/null 2> "$OUT/pmc_w.err"
python3 tools/sum_traffic.py "$(ls $OUT/pmc/*/f_counter_collection.csv $OUT/pmc/f_counter_collection.csv 2>/dev/null | head -1)" \
                             "$(ls $OUT/pmc/*/w_counter_collection.csv $OUT/pmc/w_counter_collection.csv 2>/dev/null | head -1)" "$OUT/flrelu_traffic.json"

echo "[3/6] matrix-core utilisation of the convolutions"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o m -- python3 bench.py $EAGER > /dev/null 2> "$OUT/pmc_m.err"
python3 tools/sum_mfma.py "$(ls $OUT/pmc/*/m_counter_collection.csv $OUT/pmc/m_counter_collection.csv 2>/dev/null | head -1)" > "$OUT/conv_mfma_util.txt"
rm -rf "$OUT/pmc"

echo "[4/6] config R-1024, batch 8 (the decoder of the inversion path)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/traceR" -o r -- python3 tools/time_config.py R1024 --batch 8 --iters 5 > "$OUT/configR.log" 2> "$OUT/traceR.err"
python3 tools/trace_config_r.py "$(ls $OUT/traceR/*/r_kernel_trace.csv $OUT/traceR/r_kernel_trace.csv 2>/dev/null | head -1)" > "$OUT/configR_per_layer.txt"
cp "$(ls $OUT/traceR/*/r_kernel_stats.csv $OUT/traceR/r_kernel_stats.csv 2>/dev/null | head -1)" "$OUT/configR_kernel_stats.csv"
rm -rf "$OUT/traceR"

echo "[5/6] the full bench line (traffic figure of step 2 in place)"
cp "$OUT/flrelu_traffic.json" profiles/flrelu_traffic.json
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"

echo "[6/6] done"
ls -la "$OUT"

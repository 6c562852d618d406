#!/bin/bash
# Second half of a round's profile set (the evidence files beyond tools/collect_profiles.sh), one MI355X:
#     gpurun --timeout 1150 -- 'bash tools/collect_profiles_extra.sh r03'
# Writes gpurun_out/prof_<tag>/... .  Profiler runs have the program itself after `--`; PMC passes carry --kernel-trace only.
set -eo pipefail
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
VALU="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
first() { ls "$@" 2>/dev/null | head -1; }

echo "[1/7] VALU counters of the filtered_lrelu launches, config R-1024 (radial kernels) and config T-1024"
rocprofv3 --kernel-trace --pmc $VALU --output-format csv -d "$OUT/pmcR" -o v -- python3 tools/time_config.py R1024 --batch 8 --iters 2 > "$OUT/pmcR.log" 2> "$OUT/pmcR.err"
python3 tools/sum_valu.py "$(first $OUT/pmcR/*/v_counter_collection.csv $OUT/pmcR/v_counter_collection.csv)" "$(first $OUT/pmcR/*/v_kernel_trace.csv $OUT/pmcR/v_kernel_trace.csv)" "config R-1024" > "$OUT/configR_valu_pmc.txt"
rm -rf "$OUT/pmcR"
# HBM traffic of the radial launches at the inversion batch (16 frames per forward): FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmcRf" -o f -- python3 tools/time_config.py R1024 --batch 16 --iters 2 > /dev/null 2> "$OUT/pmcRf.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmcRw" -o w -- python3 tools/time_config.py R1024 --batch 16 --iters 2 > /dev/null 2> "$OUT/pmcRw.err"
python3 tools/sum_traffic.py "$(first $OUT/pmcRf/*/f_counter_collection.csv $OUT/pmcRf/f_counter_collection.csv)" "$(first $OUT/pmcRw/*/w_counter_collection.csv $OUT/pmcRw/w_counter_collection.csv)" \
    "$OUT/flrelu_traffic_R.json" "one batch-16 R1024 forward"
rm -rf "$OUT/pmcRf" "$OUT/pmcRw"
rocprofv3 --kernel-trace --pmc $VALU --output-format csv -d "$OUT/pmcT" -o v -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-inversion --no-extras > /dev/null 2> "$OUT/pmcT.err"
python3 tools/sum_valu.py "$(first $OUT/pmcT/*/v_counter_collection.csv $OUT/pmcT/v_counter_collection.csv)" "$(first $OUT/pmcT/*/v_kernel_trace.csv $OUT/pmcT/v_kernel_trace.csv)" > "$OUT/flrelu_valu_pmc.txt"
rm -rf "$OUT/pmcT"

echo "[2/7] one PTI step, kernel by kernel"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pti" -o p -- python3 tools/time_pti.py T1024 --iters 10 > "$OUT/pti.log" 2> "$OUT/pti.err"
cp "$(first $OUT/pti/*/p_kernel_stats.csv $OUT/pti/p_kernel_stats.csv)" "$OUT/pti_step_kernel_stats.csv"
rm -rf "$OUT/pti"

echo "[3/7] the encoder forward, kernel by kernel"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc" -o e -- python3 tools/time_encoder.py --batch 16 --iters 10 > "$OUT/enc.log" 2> "$OUT/enc.err"
cp "$(first $OUT/enc/*/e_kernel_stats.csv $OUT/enc/e_kernel_stats.csv)" "$OUT/encoder_kernel_stats.csv"
rm -rf "$OUT/enc"

echo "[4/7] mixed precision per layer"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/mix" -o m -- python3 tools/time_config.py T1024 --batch 8 --iters 5 --mixed > "$OUT/mixed.log" 2> "$OUT/mix.err"
python3 tools/trace_layers.py "$(first $OUT/mix/*/m_kernel_trace.csv $OUT/mix/m_kernel_trace.csv)" > "$OUT/mixed_per_layer.txt"
rm -rf "$OUT/mix"

echo "[5/7] fp32 VALU ceiling + the MFMA-for-filters lines"
hipcc --offload-arch=gfx950 -O3 tools/microbench_valu_random.hip -o /tmp/mv 2> /dev/null
timeout -k 10 300 /tmp/mv > "$OUT/valu_ceiling.txt"

echo "[6/7] in-kernel stamps of the transform-domain convolution"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSG3_F23_STAMPS tools/f23_stamps.hip -o /tmp/f23s 2> /dev/null
for tn in 4 5 7; do echo "== TN=$tn"; timeout -k 10 120 /tmp/f23s $tn; done > "$OUT/f23_stamps.txt" 2>&1

echo "[7/7] transform-domain vs direct kernel, layer by layer (same box, same clocks)"
{ echo "direct kernel (SG3_CONV_F23 off)"; python3 tools/bench_layer.py conv L3 L5 L6 L7 L8 L9 L10 L11 --f23 off --iters 20;
  echo "transform-domain kernel forced on"; python3 tools/bench_layer.py conv L3 L5 L6 L7 L8 L9 L10 L11 --f23 on --iters 20; } > "$OUT/f23_vs_direct.txt" 2>&1
ls -la "$OUT"

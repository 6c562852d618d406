#!/bin/bash
# matrix / vector co-execution counters of the transform-domain 3x3 kernel on single layers (batch 8, T-1024 shapes)
#   bash tools/f23_pmc.sh <outdir> L6 [L5 ...]
set -e
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --output-format csv -d "$OUT/pmc" -o p -- python3 "$ROOT/tools/bench_layer.py" conv "$@" --iters 3 > "$OUT/pmc.log" 2>&1
cd "$ROOT"
python3 tools/f23_pmc_summary.py "$OUT/pmc"

#!/bin/bash
# rocprofv3 evidence for the headline kernel (run on the GPU box from the repo root):
#   tools/pmc_profile.sh [out_dir] [extra bench.py args]
# 1. --kernel-trace --stats of the bench command (average launch duration of render_kernel must agree with bench.py's roofline.kernel_ms);
# 2. four separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ counters per pass), each
#    with --kernel-trace only;
# 3. tools/pmc_summary.py -> <out_dir>/pmc_summary.json (HBM bytes per launch with the gfx950 FETCH_SIZE x2 correction, MFMA pipe
#    occupancy, effective clock).
set -e
OUT=${1:-gpurun_out/pmc}
shift || true
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-train --no-extras --cpu-rows 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- $BENCH > "$ROOT/$OUT/bench_under_trace.json" 2> "$ROOT/$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc1" -- $BENCH > /dev/null 2> "$ROOT/$OUT/pmc1.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc2" -- $BENCH > /dev/null 2> "$ROOT/$OUT/pmc2.err"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc3" -- $BENCH > /dev/null 2> "$ROOT/$OUT/pmc3.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc4" -- $BENCH > /dev/null 2> "$ROOT/$OUT/pmc4.err"
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"

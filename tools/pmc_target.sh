#!/bin/bash
# rocprofv3 evidence for one kernel family (run on the GPU box from the repo root):
#   tools/pmc_target.sh OUT_DIR KERNEL_SUBSTRING[,KERNEL_SUBSTRING...] <tools/profile_target.py arguments>
# e.g. tools/pmc_target.sh gpurun_out/pmc_v1_f16x3 render_kernel render --net v1 --mode f16x3
# One --kernel-trace --stats run + four separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; 8 SQ counters per pass; never --pmc together with a sys/hip/hsa trace).  The program itself sits
# directly after `--` (python3 <script>: no env / bash -c hop).
set -e
OUT=$1; KERNELS=$2; shift 2
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- python3 "$ROOT/tools/profile_target.py" "$@" > "$ROOT/$OUT/target_under_trace.json" 2> "$ROOT/$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc1" -- python3 "$ROOT/tools/profile_target.py" "$@" > /dev/null 2> "$ROOT/$OUT/pmc1.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc2" -- python3 "$ROOT/tools/profile_target.py" "$@" > /dev/null 2> "$ROOT/$OUT/pmc2.err"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc3" -- python3 "$ROOT/tools/profile_target.py" "$@" > /dev/null 2> "$ROOT/$OUT/pmc3.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc4" -- python3 "$ROOT/tools/profile_target.py" "$@" > /dev/null 2> "$ROOT/$OUT/pmc4.err"
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" --kernels "$KERNELS" --label "tools/profile_target.py $*" > "$OUT/pmc_summary.json"
echo "== $OUT"; python3 -c "import json,sys; d=json.load(open('$OUT/pmc_summary.json')); [print(k['kernel'][:110], json.dumps(k['derived'])) for k in d['kernels']]"

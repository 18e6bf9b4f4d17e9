#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r3d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3d_pytest.txt
python tools/bench_small_frames.py > gpurun_out/r3d_small_frames.txt 2>&1; echo "small rc=$?"
python tools/bench_configs.py --mode f16 > gpurun_out/r3d_configs_f16.txt 2>&1; echo "configs rc=$?"
python tools/trained_scene.py --net v3 > gpurun_out/r3d_tr_v3.json 2> gpurun_out/r3d_tr_v3.err; echo "v3 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v1_f16 render_kernel render --net v1 --mode f16 > gpurun_out/r3d_pmc_v1_f16.log 2>&1; echo "pmc v1 f16 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v1_bf16 render_kernel render --net v1 --mode bf16 > gpurun_out/r3d_pmc_v1_bf16.log 2>&1; echo "pmc v1 bf16 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v1_f16x3 render_kernel render --net v1 --mode f16x3 > gpurun_out/r3d_pmc_v1_f16x3.log 2>&1; echo "pmc v1 f16x3 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v2_f16 render_kernel render --net v2 --mode f16 > gpurun_out/r3d_pmc_v2_f16.log 2>&1; echo "pmc v2 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v3_f16 render_kernel render --net v3 --mode f16 > gpurun_out/r3d_pmc_v3_f16.log 2>&1; echo "pmc v3 rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_v3w_f16 render_kernel render --net v3w --mode f16 > gpurun_out/r3d_pmc_v3w_f16.log 2>&1; echo "pmc v3w rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_queue_f16 render_queue_kernel queue --net v1 --mode f16 > gpurun_out/r3d_pmc_queue_f16.log 2>&1; echo "pmc queue rc=$?"
bash tools/pmc_target.sh gpurun_out/r3d_pmc_train_bf16 train_forward_kernel,train_backward_kernel,weight_grad_kernel,weight_grad_reduce train --net v1 --mode bf16 > gpurun_out/r3d_pmc_train_bf16.log 2>&1; echo "pmc train rc=$?"
# keep only the summaries + kernel stats (the raw csv trees are large)
for d in gpurun_out/r3d_pmc_*/; do find "$d" -name "*_kernel_stats.csv" -path "*trace*" -exec cp {} "${d%/}_kernel_stats.csv" \; ; cp "$d/pmc_summary.json" "${d%/}_summary.json"; cp "$d/target_under_trace.json" "${d%/}_target.json"; done
rm -rf gpurun_out/r3d_pmc_*/
ls gpurun_out | grep r3d

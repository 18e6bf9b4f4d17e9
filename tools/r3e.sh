#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/trained_scene.py --net v3 > gpurun_out/r3e_tr_v3.json 2> gpurun_out/r3e_tr_v3.err; echo "v3 rc=$?"
for i in 1 2 3; do
  python tools/profile_target.py train --net v1 --mode bf16 --reps 25 >> gpurun_out/r3e_ab_wgrad.txt 2>/dev/null
  NRF_LIB=$PWD/nerf_few_shot_limitations_amd/libnerfhip_wgpf1.so python tools/profile_target.py train --net v1 --mode bf16 --reps 25 2>/dev/null | sed 's/^/PF1 /' >> gpurun_out/r3e_ab_wgrad.txt
  python tools/profile_target.py train --net v1 --mode bf16 --rays 16384 --samples 64 --reps 10 2>/dev/null | sed 's/^/1Mi PF2 /' >> gpurun_out/r3e_ab_wgrad.txt
  NRF_LIB=$PWD/nerf_few_shot_limitations_amd/libnerfhip_wgpf1.so python tools/profile_target.py train --net v1 --mode bf16 --rays 16384 --samples 64 --reps 10 2>/dev/null | sed 's/^/1Mi PF1 /' >> gpurun_out/r3e_ab_wgrad.txt
done
cat gpurun_out/r3e_ab_wgrad.txt
python tools/bench_small_frames.py > gpurun_out/r3e_small_frames.txt 2>&1; echo "small rc=$?"
python tools/bench_configs.py --mode f16 > gpurun_out/r3e_configs_f16.txt 2>&1; echo "configs rc=$?"
python tools/bench_configs.py --mode bf16 > gpurun_out/r3e_configs_bf16.txt 2>&1
python bench.py > gpurun_out/r3e_bench_f16_v1.json 2> gpurun_out/r3e_bench_f16_v1.err; echo "bench rc=$?"
python bench.py --net v2 > gpurun_out/r3e_bench_f16_v2.json 2> gpurun_out/r3e_bench_f16_v2.err; echo "bench v2 rc=$?"
python bench.py --net v3 > gpurun_out/r3e_bench_f16_v3.json 2> gpurun_out/r3e_bench_f16_v3.err; echo "bench v3 rc=$?"
bash tools/pmc_profile.sh gpurun_out/r3e_pmc_bench > gpurun_out/r3e_pmc_bench.log 2>&1; echo "pmc bench rc=$?"
find gpurun_out/r3e_pmc_bench -name "*_kernel_stats.csv" -path "*trace*" -exec cp {} gpurun_out/r3e_bench_kernel_stats.csv \;
cp gpurun_out/r3e_pmc_bench/pmc_summary.json gpurun_out/r3e_pmc_bench_summary.json; cp gpurun_out/r3e_pmc_bench/bench_under_trace.json gpurun_out/r3e_bench_under_trace.json
rm -rf gpurun_out/r3e_pmc_bench

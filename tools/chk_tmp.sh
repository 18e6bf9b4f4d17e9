export PYTHONPATH=$PWD
ROOT=$PWD
tools/pmc_profile.sh gpurun_out/pmc_final > gpurun_out/pmc_final.log 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/configs_trace -- python3 $ROOT/tools/bench_configs.py --mode bf16 --reps 5 > $ROOT/gpurun_out/configs_final.txt 2> $ROOT/gpurun_out/configs_final.err)
python bench.py > gpurun_out/bench_final_v1.json 2> gpurun_out/bench_final_v1.err
python bench.py --net v2 > gpurun_out/bench_final_v2.json 2> gpurun_out/bench_final_v2.err
python bench.py --net v3 > gpurun_out/bench_final_v3.json 2> gpurun_out/bench_final_v3.err
grep -v amdgpu gpurun_out/configs_final.txt | cut -c1-200
tail -c 600 gpurun_out/bench_final_v1.json

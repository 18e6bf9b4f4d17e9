#!/bin/bash
# round 3, call A: full GPU suite + trained scene + bench (f16 default) + 2-rank rehearsal + configs
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a_pytest.txt
tail -5 gpurun_out/r3a_pytest.txt
python tools/trained_scene.py --net v2 > gpurun_out/r3a_trained_v2.json 2> gpurun_out/r3a_trained_v2.err; echo "trained v2 rc=$?"
python tools/trained_scene.py --net v1 > gpurun_out/r3a_trained_v1.json 2> gpurun_out/r3a_trained_v1.err; echo "trained v1 rc=$?"
python bench.py > gpurun_out/r3a_bench_f16.json 2> gpurun_out/r3a_bench_f16.err; echo "bench rc=$?"
python bench.py --mode bf16 --no-trained-scene --no-train --cpu-rows 0 > gpurun_out/r3a_bench_bf16.json 2> gpurun_out/r3a_bench_bf16.err; echo "bench bf16 rc=$?"
python bench.py --gpus 2 --rehearse --steps 5 --warmup 1 > gpurun_out/r3a_bench_n2.json 2> gpurun_out/r3a_bench_n2.err; echo "bench n2 rc=$?"
python tools/bench_configs.py --mode f16 > gpurun_out/r3a_configs_f16.txt 2>&1; echo "configs rc=$?"
python tools/bench_configs.py --mode bf16 > gpurun_out/r3a_configs_bf16.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3a_smoke.txt 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/r3a_smoke.txt

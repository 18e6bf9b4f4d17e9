#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r3c_pytest.txt
python tools/bench_small_frames.py > gpurun_out/r3c_small_frames.txt 2>&1; echo "small rc=$?"
python tools/trained_scene.py --net v2 > gpurun_out/r3c_tr_v2.json 2> gpurun_out/r3c_tr.err; echo "v2 rc=$?"
python tools/trained_scene.py --net v1 > gpurun_out/r3c_tr_v1.json 2>> gpurun_out/r3c_tr.err; echo "v1 rc=$?"
python tools/trained_scene.py --net v2 --seed 1 > gpurun_out/r3c_tr_v2_s1.json 2>> gpurun_out/r3c_tr.err; echo "v2 s1 rc=$?"
python tools/trained_scene.py --net v1 --seed 1 > gpurun_out/r3c_tr_v1_s1.json 2>> gpurun_out/r3c_tr.err; echo "v1 s1 rc=$?"
python bench.py > gpurun_out/r3c_bench_f16.json 2> gpurun_out/r3c_bench_f16.err; echo "bench rc=$?"
python bench.py --gpus 2 --rehearse --steps 5 --warmup 1 > gpurun_out/r3c_bench_n2.json 2> gpurun_out/r3c_bench_n2.err; echo "bench n2 rc=$?"
python tools/bench_train.py > gpurun_out/r3c_bench_train.txt 2>&1; echo "train bench rc=$?"
python tools/bench_configs.py --mode f16 > gpurun_out/r3c_configs_f16.txt 2>&1; echo "configs rc=$?"

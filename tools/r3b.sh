#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3b_pytest.txt
python tools/bench_small_frames.py > gpurun_out/r3b_small_frames.txt 2>&1; echo "small rc=$?"
python tools/trained_scene.py --net v2 > gpurun_out/r3b_tr_v2.json 2> gpurun_out/r3b_tr_v2.err; echo "v2 rc=$?"
python tools/trained_scene.py --net v2 --epochs 400 > gpurun_out/r3b_tr_v2_e400.json 2>> gpurun_out/r3b_tr_v2.err; echo "v2 e400 rc=$?"
python tools/trained_scene.py --net v2 --views 20 > gpurun_out/r3b_tr_v2_v20.json 2>> gpurun_out/r3b_tr_v2.err; echo "v2 v20 rc=$?"
python tools/trained_scene.py --net v2 --train-mode f32 > gpurun_out/r3b_tr_v2_f32.json 2>> gpurun_out/r3b_tr_v2.err; echo "v2 f32 rc=$?"
python tools/trained_scene.py --net v1 > gpurun_out/r3b_tr_v1.json 2> gpurun_out/r3b_tr_v1.err; echo "v1 rc=$?"
python tools/trained_scene.py --net v1 --sigma-bias 0.5 > gpurun_out/r3b_tr_v1_sb.json 2>> gpurun_out/r3b_tr_v1.err; echo "v1 sb rc=$?"
python tools/trained_scene.py --net v1 --v1-batch 8192 > gpurun_out/r3b_tr_v1_b8k.json 2>> gpurun_out/r3b_tr_v1.err; echo "v1 b8k rc=$?"
python tools/trained_scene.py --net v1 --lr 2e-4 > gpurun_out/r3b_tr_v1_lr.json 2>> gpurun_out/r3b_tr_v1.err; echo "v1 lr rc=$?"
python tools/trained_scene.py --net v1 --train-mode f32 --epochs 120 > gpurun_out/r3b_tr_v1_f32.json 2>> gpurun_out/r3b_tr_v1.err; echo "v1 f32 rc=$?"

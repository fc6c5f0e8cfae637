#!/bin/bash
# records the round-3 numbers DESIGN.md quotes (run on the GPU box via gpurun; outputs under gpurun_out/, copied into profiles/)
set -e
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > gpurun_out/r03_g24_f32_bench.json 2> gpurun_out/r03_g24_f32_bench.err
echo "g24 f32 (+ other_configs) done"
ND_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 12 --warmup 2 --no-roofline > gpurun_out/r03_rehearsal2_bench.json 2> gpurun_out/r03_rehearsal2_bench.err
echo "rehearsal N=2 (both ranks on one GPU, gloo) done"
python tools/bench_train.py --cs 184 --batch 30 > gpurun_out/r03_train_bench_cs184_b30.json 2>/dev/null || true
python tools/bench_train.py --cs 136 --batch 30 > gpurun_out/r03_train_bench_cs136_b30.json 2>/dev/null || true
echo "train done"

# Round-3 timing experiments on the 16-bit conv_qp kernel (GPU box, via gpurun).  Needs three extra builds next to the product library:
#   make -C nind_denoise_amd/csrc ABLATE=1 BUILD=$PWD/nind_denoise_amd/csrc/build_abl OUT=$PWD/nind_denoise_amd/libnind_hip_abl.so
#   make -C nind_denoise_amd/csrc NUSE=2   BUILD=.../build_n2 OUT=.../libnind_hip_n2.so      (and NUSE=3 -> libnind_hip_n3.so)
# Output: gpurun_out/r3_abl.log (kept as profiles/r03_bf16_conv_qp_ablation.log), gpurun_out/r3_cli_stages.log
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/prof_cli_stages.py > gpurun_out/r3_cli_stages.log 2>&1 || { tail -20 gpurun_out/r3_cli_stages.log; exit 1; }
cat gpurun_out/r3_cli_stages.log | tail -3
L=nind_denoise_amd
cp $L/libnind_hip.so /tmp/real.so
LAYERS="convs1.2,convs2.2,tconvs2.0,tconvs3.0,tconvs4.0,tconvs4.2"
run() { timeout -k 10 200 python tools/bench_layers.py --dtype bf16 --batch 160 --iters 10 --layers $LAYERS 2>&1 | grep -v amdgpu.ids; }
echo "== baseline" > gpurun_out/r3_abl.log; run >> gpurun_out/r3_abl.log
cp $L/libnind_hip_abl.so $L/libnind_hip.so
for a in 0 1 2 3 4 7; do echo "== ablate build, ND_QP_ABL=$a" >> gpurun_out/r3_abl.log; ND_QP_ABL=$a run >> gpurun_out/r3_abl.log; done
cp $L/libnind_hip_n2.so $L/libnind_hip.so; echo "== NUSE=2/8" >> gpurun_out/r3_abl.log; run >> gpurun_out/r3_abl.log
cp $L/libnind_hip_n3.so $L/libnind_hip.so; echo "== NUSE=3/8" >> gpurun_out/r3_abl.log; run >> gpurun_out/r3_abl.log
cp /tmp/real.so $L/libnind_hip.so
grep -E "==|sum" gpurun_out/r3_abl.log

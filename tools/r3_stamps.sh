set -e
cd $GRAFT_REPO_ROOT
L=nind_denoise_amd
cp $L/libnind_hip.so /tmp/real.so
cp $L/libnind_hip_st.so $L/libnind_hip.so
ND_QP_DBG=128 timeout -k 10 300 python tools/bench_layers.py --dtype bf16 --batch 160 --iters 2 --layers convs1.2,convs3.2,tconvs2.0,tconvs3.0,tconvs4.0 > gpurun_out/r3_bf16_stamps.log 2>&1 || true
cp /tmp/real.so $L/libnind_hip.so
grep -v amdgpu gpurun_out/r3_bf16_stamps.log | head -60

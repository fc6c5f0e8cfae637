set -e
cd $GRAFT_REPO_ROOT
L=nind_denoise_amd
cp $L/libnind_hip.so /tmp/real.so
cp $L/libnind_hip_exp.so $L/libnind_hip.so
for cfg in "32 0" "16 0" "16 3" "32 3"; do
  set -- $cfg
  echo "== ND_WINO_OUT_NTL=$1 ND_WINO_GEMM128=$2"
  ND_WINO_OUT_NTL=$1 ND_WINO_GEMM128=$2 timeout -k 10 300 python tools/two_stream_probe.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/real.so $L/libnind_hip.so

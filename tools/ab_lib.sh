# A/B of two builds of the library on whole frames in ONE gpurun call (same box): usage  tools/ab_lib.sh <other.so>   (GPU box)
cd $GRAFT_REPO_ROOT
L=nind_denoise_amd
cp $L/libnind_hip.so /tmp/new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp $1 $L/libnind_hip.so; else cp /tmp/new.so $L/libnind_hip.so; fi
    echo "== $which"; timeout -k 10 300 python tools/ab_pool.py 2>&1 | grep -v amdgpu | grep fused
  done
done
cp /tmp/new.so $L/libnind_hip.so

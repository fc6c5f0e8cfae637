set -e
cd $GRAFT_REPO_ROOT
L=nind_denoise_amd
cp $L/libnind_hip.so /tmp/real.so
run() { timeout -k 10 200 python tools/bench_layers.py --dtype bf16 --batch 160 --iters 10 --variants 16,30 --layers convs1.2,convs3.2,tconvs2.0,tconvs3.0 2>&1 | grep -v amdgpu.ids | awk '/TF/{printf "   %-10s %s %s ms %s TF\n",$1,$6,$(NF-3),$(NF-1)} /sum/{print "   "$0}'; }
echo "== no filler"; run
for n in 8 16 32; do cp $L/libnind_hip_f$n.so $L/libnind_hip.so; echo "== $n VALU instructions after every MFMA group"; run; done
cp /tmp/real.so $L/libnind_hip.so

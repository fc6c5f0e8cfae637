# per-layer sweep of the 3x3 workgroup shapes in 16-bit storage (GPU box): auto choice (-1) against shapes 0, 1, 11, 14 of the dtype's group
cd $GRAFT_REPO_ROOT
DT=${1:-bf16}; G=${2:-15}
timeout -k 10 600 python tools/bench_layers.py --dtype $DT --batch ${3:-160} --cs ${4:-264} --iters 10 --variants=-1,$((G+0)),$((G+1)),$((G+11)),$((G+14)) 2>&1 | grep -v amdgpu.ids | grep -v "^up" | awk '/TF/{printf "%-10s %-5s %8s ms %8s TF\n",$1,$6,$(NF-3),$(NF-1)} /skipped/{print $1, $2, "skipped"} /sum/{print}'

#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> "<counters>" <python script + args...>
# One rocprofv3 counter pass (PMC + kernel trace only -- never combined with other trace domains).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CNT="$1"; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- python3 "$@"

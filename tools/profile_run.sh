#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_run.sh <tag> <bench.py args...>
# Writes gpurun_out/prof_<tag>/: rocprofv3 kernel-trace stats of `python3 bench.py <args>` and three PMC passes (MFMA busy + GUI
# active, FETCH_SIZE, WRITE_SIZE: separate passes, counters never combined with other trace domains), then the two summaries
# gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_pmc_summary.json (copy them into profiles/).
set -e
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py $* --no-cpu-baseline --no-host-leg --no-roofline --no-whole-leg --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $ARGS > $OUT.trace.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT.pmc1.log 2>&1
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT.pmc2.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT.pmc3.log 2>&1
echo "pmc write done"
cd $ROOT
DB=$(ls $OUT/trace/*/*_results.db 2>/dev/null | head -1)
[ -z "$DB" ] && DB=$(find $OUT/trace -name "*_results.db" | head -1)
python3 tools/prof_summary.py $DB --csv gpurun_out/${TAG}_kernel_stats.csv --top 16
python3 tools/pmc_summary.py $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write --command "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py $* --no-cpu-baseline --no-host-leg --no-roofline --no-whole-leg --no-other-configs" --config "$PROFILE_CONFIG" --note "three separate PMC passes; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half their bytes)" --out gpurun_out/${TAG}_pmc_summary.json
rm -rf $OUT/trace $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write
echo "summaries written"

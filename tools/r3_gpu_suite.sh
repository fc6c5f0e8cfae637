# The whole -m gpu suite, the two stack tables and the CLI wall-time record in one gpurun call.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_pytest_all.log 2>&1 || { tail -40 gpurun_out/r3_pytest_all.log; exit 1; }
tail -3 gpurun_out/r3_pytest_all.log
for dt in bf16 f32; do
  B=160; [ $dt = f32 ] && B=256
  timeout -k 10 300 python tools/stack_table.py --dtype $dt --batch $B --crop 32 > gpurun_out/r3_stack_$dt.log 2>&1
  tail -30 gpurun_out/r3_stack_$dt.log
done
timeout -k 10 600 python tools/record_cli_wall.py > gpurun_out/r03_cli_wall.json 2> gpurun_out/r03_cli_wall.err || { tail -30 gpurun_out/r03_cli_wall.err; exit 1; }
grep wall_s gpurun_out/r03_cli_wall.json

set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "stream or sharded" > gpurun_out/r3_pytest_stream.log 2>&1 || { tail -30 gpurun_out/r3_pytest_stream.log; exit 1; }
tail -3 gpurun_out/r3_pytest_stream.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err || { tail -30 gpurun_out/r3_bench_a.err; exit 1; }
tail -5 gpurun_out/r3_bench_a.err
ND_BENCH_REHEARSAL=1 timeout -k 10 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-roofline > gpurun_out/r3_bench_reh2.json 2> gpurun_out/r3_bench_reh2.err || { tail -30 gpurun_out/r3_bench_reh2.err; exit 1; }
tail -3 gpurun_out/r3_bench_reh2.err
echo done

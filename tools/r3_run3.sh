set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_host_abi.py -x -q -m gpu -k "worker or cli or thin" > gpurun_out/r3_pytest_worker.log 2>&1 || { tail -40 gpurun_out/r3_pytest_worker.log; exit 1; }
tail -3 gpurun_out/r3_pytest_worker.log
timeout -k 10 600 python tools/record_cli_wall.py > gpurun_out/r03_cli_wall.json 2> gpurun_out/r03_cli_wall.err || { tail -30 gpurun_out/r03_cli_wall.err; exit 1; }
cat gpurun_out/r03_cli_wall.json

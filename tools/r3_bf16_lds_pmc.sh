# Round-3 diagnostics (GPU box, via gpurun): the bf16 MFMA shape micro-benchmark and two LDS / wait counter passes over the bf16 bench.
# Output: gpurun_out/r3_ubench_bf16_shape.log, gpurun_out/r3_bf16_lds_pmc.json (kept under profiles/r03_*)
set -e
R=$GRAFT_REPO_ROOT
cd $R/tools/ubench && timeout -k 10 300 ./mfma_bf16_shape > $R/gpurun_out/r3_ubench_bf16_shape.log 2>&1
cd $R
A="$R/bench.py --dtype bf16 --batch 160 --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg --no-roofline --no-whole-leg"
timeout -k 10 400 tools/pmc_pass.sh r3_pmc_lds1 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $A > gpurun_out/r3_pmc_lds1.log 2>&1
timeout -k 10 400 tools/pmc_pass.sh r3_pmc_lds2 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" $A > gpurun_out/r3_pmc_lds2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r3_pmc_lds1 gpurun_out/r3_pmc_lds2 --out gpurun_out/r3_bf16_lds_pmc.json
rm -rf gpurun_out/r3_pmc_lds1 gpurun_out/r3_pmc_lds2
echo done

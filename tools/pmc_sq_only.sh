export TMPDIR=/tmp
R=$(pwd)
PMC="--steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events --no-extras"
cd /tmp
(cd "$R" && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o sq -- python3 bench.py $PMC > /dev/null 2>&1
 python3 tools/pmc_sq.py /tmp/pmc_sq gpurun_out/sq_counters_patch_kernels.txt > /dev/null)

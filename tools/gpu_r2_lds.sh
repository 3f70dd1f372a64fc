# Baseline timing + one PMC pass with the LDS counters (is the small-scene walk bound by LDS bank conflicts?).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/time_workloads.py > gpurun_out/time_r02b.txt 2>&1 || { tail -20 gpurun_out/time_r02b.txt; exit 1; }
cat gpurun_out/time_r02b.txt
bash tools/gpu_pmc.sh r02_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" > gpurun_out/pmc_r02_lds.txt 2>&1 || { tail -20 gpurun_out/pmc_r02_lds.txt; exit 1; }
grep -A 9 -E "k_extend|k_occlude_guided|k_shade_guided" gpurun_out/pmc_r02_lds.txt

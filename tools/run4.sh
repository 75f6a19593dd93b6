#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4d
cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/probe_mfma_f64_sustained.bin 2>&1 | tee gpurun_out/r4d/mfma_f64.log
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -c1-18 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_dgemm_ex.py 5 9984 > $GRAFT_REPO_ROOT/gpurun_out/r4d/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done

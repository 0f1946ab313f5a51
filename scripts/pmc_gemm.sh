#!/bin/bash
# PMC passes for one GEMM shape (separate rocprofv3 runs per counter group, --pmc only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE="$1"; TAG="$2"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_${TAG}/$n -- python scripts/bench_gemm.py "$SHAPE" 2 > gpurun_out/pmc_${TAG}_$n.log 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc_${TAG}

#!/bin/bash
# Memory-path PMC of two launches of the SAME product (proj: M = 63 744, N = K = 768): the forward with the bias + fp32-residual epilogue
# (kind 3, phased kernel) and the data gradient with the bias -> bf16 epilogue (kind 1) -- what the TA / TCP / TCC queues say about the
# fp32 epilogue's ~11 B/clk/CU.  Separate rocprofv3 passes per group, --pmc only.
#   bash scripts/pmc_mempath.sh [filter]      -> gpurun_out/mempath_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="${1:-NT proj}"
O=gpurun_out/pmc_mempath; rm -rf $O; mkdir -p $O
i=0
# (the TA_* groups -- TA_TA_BUSY_sum, TA_*_STALLED_BY_*, TA_BUFFER_* / TA_FLAT_* wavefronts -- and the TCP_LFIFO / RFIFO / TAGCONFLICT group HANG
# rocprofv3 on this stack until the timeout kills it: not requested)
for grp in "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA_RDREQ_sum" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i + 1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 scripts/bench_gemm.py "$F" 2 > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $O/g$i.log; }
done
python3 scripts/pmc_summary.py $O gemm > gpurun_out/mempath_pmc.txt 2>&1
rm -rf $O
cat gpurun_out/mempath_pmc.txt

#!/bin/bash
# One GPU-box session that refreshes every measured artefact of the round under gpurun_out/r02/ (copied into profiles/ afterwards):
#   bench line, rocprofv3 kernel stats of the bench, HBM traffic (PMC, separate passes) with the source stamp, per-kernel PMC for the
#   dominant GEMM kernels, the per-block GEMM table and the hipBLASLt bar.   usage: bash scripts/profile_round.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 bench.py --steps 5 --warmup 2 --no_cpu_baseline > $O/ktrace.log 2>&1 || exit 1
cp $(ls $O/ktrace/*/*kernel_stats.csv | head -1) $O/bench_vitb_b128_kernel_stats.csv 2>/dev/null
echo "kernel trace done"
rm -rf gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic/$c -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline > $O/traffic_$c.log 2>&1 || exit 1
done
python3 scripts/traffic_summary.py gpurun_out/traffic $O/gemm_traffic.json > $O/gemm_traffic_summary.txt 2>&1
echo "traffic done"; cat $O/gemm_traffic_summary.txt | cut -c1-400
cp $O/gemm_traffic.json profiles/r02_gemm_traffic.json   # (the box-local copy: the bench line below embeds the traffic of THIS tree)
python3 bench.py --steps 20 --warmup 5 > $O/bench_vitb_b128_line.json 2> $O/bench_vitb_b128.err || exit 1
echo "bench done"; tail -c 600 $O/bench_vitb_b128_line.json
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_r02/$n -- python3 scripts/bench_gemm.py "N" 2 > $O/pmc_$n.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_r02 > $O/gemm_pmc.txt 2>&1
echo "pmc done"
python3 scripts/bench_gemm.py > $O/bench_gemm_block.txt 2>&1
python3 scripts/bench_gemm_torch.py > $O/bench_gemm_hipblaslt.txt 2>&1
python3 scripts/bench_attn.py > $O/bench_attn.txt 2>&1; python3 scripts/bench_attn.py 501 >> $O/bench_attn.txt 2>&1
for f in $O/bench_gemm_block.txt $O/bench_gemm_hipblaslt.txt $O/bench_attn.txt; do tail -n 3 $f; done

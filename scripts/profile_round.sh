#!/bin/bash
# One GPU-box session that refreshes the measured artefacts of the round under gpurun_out/<round>/ (copied into profiles/ afterwards):
#   per workload: rocprofv3 kernel stats of the bench, HBM traffic of its GEMM launches (PMC, separate FETCH_SIZE / WRITE_SIZE passes,
#   stamped with the kernel-source hash) and the bench line (which embeds that traffic); for the headline workload also the per-kernel
#   PMC of the GEMM families, the per-block GEMM table, the hipBLASLt bar and the attention table.
#   usage: bash scripts/profile_round.sh [workload ...]        (default: all four)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${SA_ROUND:-r05}                      # file-name prefix of the round's artefacts (profiles/${R}_*)
O=gpurun_out/$R; mkdir -p $O
WL=${@:-vit_base_bt_10s vit_tiny_bt_10s vit_base_byol_10s vit_large_mae_10s}
if [ $# -eq 0 ]; then rm -f $O/gemm_traffic.json; else cp profiles/${R}_gemm_traffic.json $O/gemm_traffic.json; fi   # named workloads: refresh their entries only
for w in $WL; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no_cpu_baseline > $O/ktrace_$w.log 2>&1 || exit 1
  cp $(ls $O/ktrace_$w/*/*kernel_stats.csv | head -1) $O/bench_${w}_kernel_stats.csv 2>/dev/null
  rm -rf $O/ktrace_$w
  echo "kernel trace $w done"
  rm -rf gpurun_out/traffic
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic/$c -- python3 bench.py --workload $w --steps 2 --warmup 1 --profile_steps 1 --no_cpu_baseline > $O/traffic_${w}_$c.log 2>&1 || exit 1
  done
  python3 scripts/traffic_summary.py gpurun_out/traffic $O/gemm_traffic.json $w > $O/gemm_traffic_summary_$w.txt 2>&1
  rm -rf gpurun_out/traffic
  echo "traffic $w done"; cut -c1-300 $O/gemm_traffic_summary_$w.txt
  cp $O/gemm_traffic.json profiles/${R}_gemm_traffic.json   # (the box-local copy: the bench line below embeds the traffic of THIS tree)
  if [ $w = vit_base_bt_10s ]; then
    python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_${w}_line.json 2> $O/bench_$w.err || exit 1
  else
    python3 bench.py --workload $w --steps 10 --warmup 3 --no_cpu_baseline > $O/bench_${w}_line.json 2> $O/bench_$w.err || exit 1
  fi
  echo "bench $w done"; cut -c1-330 $O/bench_${w}_line.json
done
case " $WL " in *" vit_base_bt_10s "*) ;; *) exit 0;; esac
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$R/$n -- python3 scripts/bench_gemm.py "N" 2 > $O/pmc_$n.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_$R > $O/gemm_pmc.txt 2>&1
rm -rf gpurun_out/pmc_$R
echo "pmc done"
python3 scripts/bench_gemm.py > $O/bench_gemm_block.txt 2>&1
python3 scripts/bench_gemm_torch.py > $O/bench_gemm_hipblaslt.txt 2>&1
python3 scripts/bench_attn.py > $O/bench_attn.txt 2>&1; python3 scripts/bench_attn.py 501 >> $O/bench_attn.txt 2>&1
for f in $O/bench_gemm_block.txt $O/bench_gemm_hipblaslt.txt $O/bench_attn.txt; do tail -n 3 $f; done
bash scripts/pmc_attn.sh > /dev/null 2>&1; cp gpurun_out/attn_pmc.txt $O/attn_pmc.txt 2>/dev/null; head -n 12 $O/attn_pmc.txt   # (profiles/${R}_attn_pmc.txt)
python3 scripts/bench_frontend.py > $O/bench_frontend.txt 2>&1; tail -n 8 $O/bench_frontend.txt
bash scripts/pmc_frontend.sh $O/pmc_fe > $O/frontend_pmc.txt 2>&1; rm -rf $O/pmc_fe; head -n 3 $O/frontend_pmc.txt     # (profiles/${R}_frontend_pmc.txt = this + a reading)
# BASELINE config 3's global batch on ONE GPU (strong-scaling anchor, ~200 GB resident), and the headline through the self-launch path
python3 bench.py --global_batch 1024 --steps 5 --warmup 2 --no_cpu_baseline > $O/bench_vit_base_bt_10s_b1024_line.json 2> $O/bench_b1024.err || echo "b1024 failed"
cut -c1-200 $O/bench_vit_base_bt_10s_b1024_line.json
SA_BENCH_SELF_LAUNCH=1 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline > $O/bench_self_launch_line.json 2> $O/bench_self_launch.err || echo "self-launch failed"
cut -c1-200 $O/bench_self_launch_line.json

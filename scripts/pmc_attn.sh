cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_attn
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_attn/$n -- python3 scripts/bench_attn.py > gpurun_out/pmc_attn_$n.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_attn attn > gpurun_out/attn_pmc.txt 2>&1
grep -A40 "attn_" gpurun_out/attn_pmc.txt | cut -c1-100

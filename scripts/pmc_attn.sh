#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_attn/$n -- python scripts/bench_attn.py > gpurun_out/pmc_attn_$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY

"""HBM bytes per GEMM launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, KiB units; FETCH doubled per the gfx950 note in
MI355X_MICROARCH.md), overall and per device kernel (same family names as ops.gemm_kernel_family / bench.py's roofline object)."""
import csv, glob, json, sys, collections
root, out = sys.argv[1], sys.argv[2]


def family(name):
    if "colsum" in name or "gemm" not in name or "stream_reduce" in name:      # (reduce launches: not GEMM kernels; "sagemm::" is in their signature)
        return None
    if "gemm_tn_stream_kernel" in name: return "gemm_tn_stream_kernel"
    if "gemm_tn_stream256_kernel" in name: return "gemm_tn_stream256_kernel"
    if "gemm256_phase_kernel" in name: return "gemm256_phase_kernel"
    if "gemm256_persist_kernel" in name: return "gemm256_persist_kernel"
    if "gemm256_ring_kernel" in name: return "gemm256_ring_kernel<split-K>" if "true>(" in name.split("gemm256_ring_kernel")[1][:24].replace(" ", "").split(",")[-1] else "gemm256_ring_kernel"
    if "gemm256_kernel" in name: return "gemm256_kernel<split-K>"
    if "gemm_kernel" in name: return "gemm_kernel<split-K>" if name.split("gemm_kernel<")[1].split(">")[0].replace(" ", "").endswith("false") else "gemm_kernel"
    return "other"


tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        for key in (fam, "__all__"):
            t = tot[key][r["Counter_Name"]]
            t[0] += float(r["Counter_Value"]); t[1] += 1


def summarise(d):
    fetch_kb, nf = d["FETCH_SIZE"]; write_kb, nw = d["WRITE_SIZE"]
    return {"launches": nf, "fetch_bytes_per_launch_raw": fetch_kb * 1024 / max(nf, 1),
            "fetch_bytes_per_launch_corrected_x2": 2 * fetch_kb * 1024 / max(nf, 1),
            "write_bytes_per_launch": write_kb * 1024 / max(nw, 1),
            "hbm_bytes_per_launch": (2 * fetch_kb / max(nf, 1) + write_kb / max(nw, 1)) * 1024}


res = summarise(tot["__all__"])
res["per_kernel"] = {k: summarise(v) for k, v in tot.items() if k != "__all__"}
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
res["source_sha"] = bench.source_sha()          # bench.py ignores this file once the kernel sources differ
res["note"] = "sa_gemm_bf16 launches of `bench.py --steps 2 --warmup 1 --profile_steps 1` (5 steps); FETCH_SIZE doubled per the gfx950 correction"
workload = sys.argv[3] if len(sys.argv) > 3 else "vit_base_bt_10s"
# one file for all workloads: {"source_sha": ..., "workloads": {name: summary}}; an entry collected at other kernel sources is dropped
import os
allw = {}
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if old.get("source_sha") == res["source_sha"]:
            allw = old.get("workloads", {})
    except Exception:
        allw = {}
allw[workload] = {k: v for k, v in res.items() if k not in ("source_sha",)}
json.dump({"source_sha": res["source_sha"], "workloads": allw}, open(out, "w"), indent=1)
print(json.dumps({k: (v if k != "per_kernel" else {n: round(x["hbm_bytes_per_launch"] / 1e6, 1) for n, x in v.items()}) for k, v in res.items()}))

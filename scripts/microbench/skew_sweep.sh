mkdir -p gpurun_out/r03
for dyn in 0 1; do for ns in 0 8000 16000 24000 36000; do echo "== dyn=$dyn skew_ns=$ns"; SA_GEMM_DYNAMIC=$dyn SA_GEMM_SKEW_NS=$ns timeout -k 10 120 python scripts/bench_gemm.py "" 10 2>&1 | grep -v amdgpu.ids | grep -E "fwd|dgrad|ALL"; done; done > gpurun_out/r03/skew.txt 2>&1
cat gpurun_out/r03/skew.txt

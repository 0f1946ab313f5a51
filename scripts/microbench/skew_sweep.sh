# Record of the start-skew experiment (second pass): SA_GEMM_SKEW_NS delayed the first tile of workgroup b of an XCD by
# skew * (b >> 3) / (grid / 8) in the persistent kernels.  It bought nothing and the knob is no longer in the library: this sweep now
# only repeats the unskewed measurement.
mkdir -p gpurun_out/r03
for dyn in 0 1; do for ns in 0 8000 16000 24000 36000; do echo "== dyn=$dyn skew_ns=$ns"; SA_GEMM_DYNAMIC=$dyn SA_GEMM_SKEW_NS=$ns timeout -k 10 120 python scripts/bench_gemm.py "" 10 2>&1 | grep -v amdgpu.ids | grep -E "fwd|dgrad|ALL"; done; done > gpurun_out/r03/skew.txt 2>&1
cat gpurun_out/r03/skew.txt

for d in 0 1 2 3 4 6; do echo "STAGGER=$d"; SA_GEMM_STAGGER=$d timeout -k 10 100 python scripts/bench_gemm.py 2>&1 | grep -E "fwd |dgrad|ALL" ; done

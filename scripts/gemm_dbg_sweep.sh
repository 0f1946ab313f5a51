for d in ${@:-0 16}; do echo "DBG=$d"; SA_GEMM_DBG=$d timeout -k 10 100 python scripts/bench_gemm.py 2>&1 | grep "fwd " ; done

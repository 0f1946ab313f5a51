for w in vit_tiny_bt_10s vit_base_byol_10s vit_large_mae_10s; do timeout -k 10 300 python3 bench.py --workload $w --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['config']['workload'], d['value'], 'clips/s', d['ms_per_step'], 'ms', 'gemm', d['roofline']['all_gemm']['achieved'], 'TF/s', 'whole', d['roofline'].get('whole_step_tflops'))" || exit 1; done

for m in 0 1 2; do
NC_C8X=$m timeout 600 python bench.py --crop 148 --batch 4 --precision bf16 --workload train --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('NC_C8X=$m ms_per_step %.2f' % j['ms_per_step'], {k: (v['ms_per_step'], v['tflops']) for k, v in j['roofline']['classes'].items()})
"
done

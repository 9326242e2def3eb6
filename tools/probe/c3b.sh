for v in base bd4 a64; do
if [ $v = base ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$PWD/neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_$v.so; fi
timeout 600 python bench.py --crop 148 --batch 4 --precision bf16 --workload train --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
c = j['roofline']['classes']
print('$v ms_per_step %.2f' % j['ms_per_step'], {k: (v['ms_per_step'], v['tflops']) for k, v in c.items() if '_lp_k3' in k or '_lp_k5' in k})
"
done

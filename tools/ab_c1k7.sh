# same-box alternation: Conv3d(1,64,7) forward + data gradient on the two-term kernels (default) vs the fp32 matrix kernels (NC_C1K7_H2=0)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_c1k7.py tests/test_gpu_structured.py tests/test_gpu_h2.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  NC_C1K7_H2=$v python3 bench.py --workload train --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=j['roofline']['classes']
print('NC_C1K7_H2=$v ms_per_step %.3f' % j['ms_per_step'], {k:c[k]['ms_per_step'] for k in c if 'k7' in k})"
done

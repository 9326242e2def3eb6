cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --no-cpu-baseline > $O/bench_config3.json 2>/dev/null
timeout 600 python3 bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_athena.json 2>/dev/null
for v in 2 1 2 1; do NC_DL_COLLAPSE=$v timeout 600 python3 bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('athena NC_DL_COLLAPSE=$v ms_per_step %.3f' % j['ms_per_step'])"; done
python3 - <<'PY'
import json
for f in ('bench_config3','bench_athena'):
    j=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'ms_per_step %.3f'%j['ms_per_step'], j['roofline']['kernel_class'], j['roofline']['frac'])
PY

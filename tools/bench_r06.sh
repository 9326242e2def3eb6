cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --no-cpu-baseline > $O/bench_config3.json 2>/dev/null
timeout 600 python3 bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_athena.json 2>/dev/null
python3 - <<'PY'
import json
for f in ('bench_default','bench_config3','bench_athena'):
    try:
        j=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1])
        print(f, 'ms_per_step %.3f'%j['ms_per_step'], 'frac', j.get('roofline',{}).get('frac'), 'infer', j.get('inference',{}).get('seconds_per_volume'), (j.get('inference',{}).get('roofline') or {}).get('frac'), 'parity', (j.get('parity_vs_cpu_oracle') or {}).get('ok'))
    except Exception as e: print(f, 'ERR', e)
PY

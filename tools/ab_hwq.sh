cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do for q in 1 0; do
export NC_ATHENA_D_EARLY=$q
python3 bench.py --workload train --model athena --data structured --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('athena NC_ATHENA_D_EARLY=$q ms', round(j['ms_per_step'],3), {k: round(v,5) for k,v in list(j['config']['first_step_losses'].items())[:4]}, {k: v for k,v in list(j['config']['losses'].items())[:4]})"
done; done

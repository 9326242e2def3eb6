# usage: tools/ab_generic.sh VAR v1 v2 ...   -- same-box alternation of an environment switch on the train step (10 steps after 3)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@" "$@"; do
  env $VAR=$v python3 bench.py --workload train --no-cpu-baseline --steps 10 --warmup 3 --no-prof 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v ms_per_step %.3f' % j['ms_per_step'], {k: round(v, 5) for k, v in list(j['config']['first_step_losses'].items())[:3]})"
done

# same-box alternation of (cubes per call, streams) for the diced inference (480^3: 125 cubes)
cd $GRAFT_REPO_ROOT
for cfg in "1 3" "3 1" "3 2" "3 3" "1 3" "3 2" "5 1" "5 2" "2 3" "6 1"; do
  set -- $cfg
  NC_INFER_BATCH=$1 NC_INFER_STREAMS=$2 python3 bench.py --workload infer --volume 480 --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 streams $2: s/volume', round(j['ms_per_step']/1e3,4), j['config'].get('seconds_per_volume'))"
done

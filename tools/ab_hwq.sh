# NC_D_NSTREAMS: side streams the discriminator jobs are dealt onto (the runtime maps all streams onto 4 hardware queues)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do for q in 2 3 6; do
export NC_D_NSTREAMS=$q
python3 bench.py --workload train --model athena --data structured --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('athena NC_D_NSTREAMS=$q ms', round(j['ms_per_step'],3))"
done; for q in 2 3 4; do
export NC_D_NSTREAMS=$q
python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('apollo NC_D_NSTREAMS=$q ms', round(j['ms_per_step'],3))"
done; done

# GPU_MAX_HW_QUEUES (HIP runtime: hardware queues the process's streams are mapped onto; default 4): the Athena step runs six discriminator
# streams beside the main one, the Apollo step four
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do for q in 5 6 4; do
export GPU_MAX_HW_QUEUES=$q
python3 bench.py --workload train --model athena --data structured --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('athena GPU_MAX_HW_QUEUES=$q ms', round(j['ms_per_step'],3))"
python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('apollo GPU_MAX_HW_QUEUES=$q ms', round(j['ms_per_step'],3))"
done; done

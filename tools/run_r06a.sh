cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_c8.py -x -q -k "oracle or fallback or survives or follows_its_forward or launches_its_own" > $O/tests_a.log 2>&1
tail -5 $O/tests_a.log
timeout 900 python bench.py > $O/bench_start.json 2> $O/bench_start.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_start.json').read().strip().splitlines()[-1])
print('ms_per_step', j['ms_per_step'], 'infer s', j['inference']['seconds_per_volume'])
print(json.dumps(j.get('parity_vs_cpu_oracle'), indent=1))
PY

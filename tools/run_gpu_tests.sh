cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1
tail -15 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py --workload train --no-cpu-baseline > $O/bench_train.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_train.json').read().strip().splitlines()[-1])
print('train ms_per_step', j['ms_per_step'], 'layer-by-layer', j['deep_linear_tail']['layer_by_layer']['ms_per_step'])
for k,v in j['roofline']['classes'].items(): print(k, v)
PY

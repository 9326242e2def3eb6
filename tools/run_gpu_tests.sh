cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1
tail -8 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_default.json').read().strip().splitlines()[-1])
print('ms_per_step %.3f'%j['ms_per_step'], 'frac', j['roofline']['frac'], 'infer', j['inference']['seconds_per_volume'], j['inference']['roofline']['frac'], 'parity', j['parity_vs_cpu_oracle'])
print('layered', j['deep_linear_tail'].get('layer_by_layer'))
PY

#!/bin/bash
# same-box A/B of the 900^3 diced inference: NC_SPLIT_TERMS given as arguments, alternating
for m in "$@"; do
  echo "infer 900^3 NC_SPLIT_TERMS=$m $(NC_SPLIT_TERMS=$m timeout 900 python3 bench.py --workload infer --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); i=j.get('inference', j); print('s_per_volume %.3f' % i['seconds_per_volume'], 'value %.3e' % i['value'], i.get('roofline',{}).get('whole_volume_tflops'))")"
done

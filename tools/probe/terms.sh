#!/bin/bash
# same-box A/B of the training step: NC_SPLIT_TERMS values as arguments, alternating; extra bench flags in $FLAGS
for m in "$@"; do
  echo "train $FLAGS NC_SPLIT_TERMS=$m $(NC_SPLIT_TERMS=$m timeout 600 python3 bench.py --workload train --no-cpu-baseline --steps 8 --warmup 3 $FLAGS 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'], ' '.join('%s %.5f' % (k, v) for k, v in list(j['config']['first_step_losses'].items())[:4]))")"
done

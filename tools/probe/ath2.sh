#!/bin/bash
# same-box A/B of the Athena step: NC_P2D masks given as arguments, alternating
for m in "$@"; do
  echo "athena 108^3 structured NC_P2D=$m $(NC_P2D=$m timeout 600 python3 bench.py --workload train --model athena --data structured --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'], 'first-step G_A %.5f D_B_xz %.5f' % (j['config']['first_step_losses']['G_A'], j['config']['first_step_losses']['D_B_xz']))")"
done

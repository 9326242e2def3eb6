#!/bin/bash
# A/B of the positions threshold of the split-operand PatchGAN kernels (NC_P2D_MIN) on the Apollo steps
run() { python3 bench.py --workload train --no-cpu-baseline --steps 8 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'], 'G_A %.5f D_A_lateral %.5f' % (j['config']['first_step_losses']['G_A'], j['config']['first_step_losses']['D_A_lateral']))"; }
for m in 8192 256 8192 256; do echo "apollo 108 fp32 NC_P2D_MIN=$m $(NC_P2D_MIN=$m run)"; done
for m in 8192 256 8192 256; do echo "apollo 148x4 bf16 NC_P2D_MIN=$m $(NC_P2D_MIN=$m run --crop 148 --batch 4 --precision bf16)"; done

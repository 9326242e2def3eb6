# Same-box A/B of the round-3 switches (each line: one bench.py run with one switch flipped).  Output: gpurun_out/ab_r03.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_r03.txt; : > $O
run() { # label, env assignment, bench args
  env $2 python3 bench.py $3 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %-22s ms_per_step %.2f' % ('$1', '$2', d['ms_per_step']))" >> $O
}
for rep in 1 2; do
  run "train 108^3"        "NC_X=0"            "--workload train --steps 8 --warmup 3"
  run "train 108^3"        "NC_S3X_NCB7=0"     "--workload train --steps 8 --warmup 3"
  run "train 108^3"        "NC_S3X_WGRAD=0"    "--workload train --steps 8 --warmup 3"
  run "train 108^3"        "NC_CONVT_S3X=0"    "--workload train --steps 8 --warmup 3"
  run "train 108^3"        "NC_S3X=0"          "--workload train --steps 8 --warmup 3"
  run "configs[3] 4x148^3 bf16" "NC_X=0"       "--workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2"
  run "configs[3] 4x148^3 bf16" "NC_HX_WGRAD=0" "--workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2"
  run "infer 480^3"        "NC_X=0"            "--workload infer --volume 480 --steps 2 --warmup 1"
  run "infer 480^3"        "NC_CONVT_S3X=0"    "--workload infer --volume 480 --steps 2 --warmup 1"
  run "infer 480^3"        "NC_C1K3=0"         "--workload infer --volume 480 --steps 2 --warmup 1"
  run "infer 480^3"        "NC_S3X_NCB7=0"     "--workload infer --volume 480 --steps 2 --warmup 1"
done
cat $O

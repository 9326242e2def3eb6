#!/bin/bash
# HBM bytes per launch of the split-operand kernel classes (the default train command's own step count) and per 140^3 cube of a 480^3 diced
# inference, on the final round-5 tree: separate --pmc passes, each under its own timeout.  Outputs under gpurun_out/r05pmc.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05pmc
rm -rf $O; mkdir -p $O
T="timeout 700"
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/tf.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/pmc_aggregate.py $O/tf $O/tw $O/train.json > /dev/null
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/if -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/if.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/iw -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/iw.log 2>&1
python3 tools/pmc_cube.py $O/if $O/iw > $O/cube.json
rm -rf $O/tf $O/tw $O/if $O/iw
cat $O/train.json | head -80; cat $O/cube.json

#!/bin/bash
# HBM bytes per launch of the split-operand kernel classes (train step) and per 140^3 cube of the inference with the two-term form (the
# default): separate --pmc passes, each under its own timeout (tools/pmc_split.sh of round 2/3, re-run on the final round-4 tree)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
T="timeout 500"
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/tf.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/pmc_aggregate.py $O/tf $O/tw $O/train.json > /dev/null
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/if -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/if.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/iw -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/iw.log 2>&1
python3 tools/pmc_cube.py $O/if $O/iw 54 > $O/cube.json
rm -rf $O/tf $O/tw $O/if $O/iw
cat $O/train.json | head -60; cat $O/cube.json

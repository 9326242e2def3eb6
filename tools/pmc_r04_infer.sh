#!/bin/bash
# the inference half of tools/pmc_r04.sh alone (HBM bytes per 140^3 cube), each pass under its own timeout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
T="timeout 500"
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/if -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/if.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/iw -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/iw.log 2>&1
python3 tools/pmc_cube.py $O/if $O/iw 54 > $O/cube.json
rm -rf $O/if $O/iw
cat $O/cube.json

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/tw
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/trace_chain.py $O/tw 1 > $O/train108_chain1.txt 2>&1
python3 tools/trace_chain.py $O/tw 2 > $O/train108_chain2.txt 2>&1
rm -rf $O/tw
head -130 $O/train108_chain1.txt

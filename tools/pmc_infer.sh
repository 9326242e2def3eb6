# HBM bytes per 140^3 cube forward of the diced inference (all kernels): two --pmc passes over a 300^3 volume (27 cubes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02/pmc_infer
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline > $O/w.log 2>&1
rm -f $O/*/t_kernel_trace.csv $O/*/t_agent_info.csv
ls -la $O/*

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/tw
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --model athena --data structured --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/trace_chain.py $O/tw 1 > $O/athena_chain1.txt 2>&1
rm -rf $O/tw
python3 - <<'PY'
import re
L=open('gpurun_out/r06/athena_chain1.txt').read().splitlines()
print(L[0])
gaps=[]; 
for l in L[1:]:
    m=re.match(r'\s*\+\s*([\d.]+) us\s+gap\s+([-\d.]+)\s+(\S+.*?)\s+([\d.]+) us$', l)
    if m: gaps.append((float(m.group(1)), float(m.group(2)), m.group(3).strip(), float(m.group(4))))
tot_gap=sum(g for _,g,_,_ in gaps if g>0); busy=sum(d for *_,d in gaps)
print('kernels', len(gaps), 'busy %.0f us, gaps %.0f us'%(busy, tot_gap))
big=[x for x in gaps if x[1]>100]
print('gaps > 100 us:', len(big), 'sum %.0f us'%sum(x[1] for x in big))
for x in big[:40]: print('  at +%.0f us gap %.0f before %s (%.0f us)'%x)
PY

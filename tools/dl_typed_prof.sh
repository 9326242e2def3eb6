cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; rm -rf $O/dlp; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dlp -o t -- python3 tools/dl_typed_prof.py 2 > $O/dlp.log 2>&1
find $O/dlp -name "*kernel_stats.csv" -exec cp {} $O/dl_typed_kernel_stats.csv \;
rm -rf $O/dlp
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06/dl_typed_kernel_stats.csv')))
tot=0
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:28]:
    print('%8.1f us/iter  n/iter %4.1f  %s'%(float(r['TotalDurationNs'])/6/1e3,int(r['Calls'])/6,r['Name'][:100]))
    tot+=float(r['TotalDurationNs'])
print('total us/iter', sum(float(r['TotalDurationNs']) for r in rows)/6/1e3)
PY

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O; rm -rf $O/sh_sq $O/sh_lds
T="timeout 300"
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sh_sq -o t -- python3 tools/w64_shapes.py > $O/sh_sq.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/sh_lds -o t -- python3 tools/w64_shapes.py > $O/sh_lds.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06w'
for sub in ('sh_sq','sh_lds'):
    disp=collections.OrderedDict()
    for f in glob.glob(O+'/'+sub+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_conv_s3' not in r['Kernel_Name']: continue
            k=int(r['Dispatch_Id'])
            e=disp.setdefault(k,dict(name=r['Kernel_Name'][30:60],dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,c={}))
            e['c'][r['Counter_Name']]=e['c'].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
    for k,e in sorted(disp.items()):
        if e['dur']<200: continue
        c=e['c']
        if sub=='sh_sq':
            gui=c['GRBM_GUI_ACTIVE']/8
            print(sub,k,e['name'],'%.0f us clk %.2f busy %.3f wait_any %.3f'%(e['dur'],gui/(e['dur']*1e3),c['SQ_VALU_MFMA_BUSY_CYCLES']/(gui*1024),c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']))
        else:
            print(sub,k,e['name'],'%.0f us bank_conflict/idx_active %.3f  lds_insts %.3g wait_lds %.3f'%(e['dur'],c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1),c['SQ_INSTS_LDS'],c['SQ_WAIT_INST_LDS']/c['SQ_WAVE_CYCLES']))
PY
rm -rf $O/sh_sq $O/sh_lds

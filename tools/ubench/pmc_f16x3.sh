cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 2 3 7 8; do
rm -rf /tmp/pm_$v
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pm_$v -o p -- $R/tools/ubench/ubench_f16x3 49152 0 $v > /dev/null 2>&1
python3 - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob('/tmp/pm_$v/**/*_results.db',recursive=True)[0])
rows={}
for name,disp,c,val in db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by name, dispatch_id, counter_name"):
    rows.setdefault(c,[]).append(val)
out={c:sum(v)/len(v) for c,v in rows.items()}
g=out.get('GRBM_GUI_ACTIVE',0)/8
print('variant $v', {k:round(v) for k,v in out.items()}, 'mfma_util %.3f'%(out.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g*1024) if g else 0), 'lds conflict frac %.3f'%(out.get('SQ_LDS_BANK_CONFLICT',0)/max(1,out.get('SQ_LDS_IDX_ACTIVE',1))))
PY
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lds; mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/a -- $R/tools/bin/kbench_a0 512 10000 512 > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/b -- $R/tools/bin/kbench_a0 512 10000 512 > $O/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_lds'
for d in ('a','b'):
    acc=collections.defaultdict(list)
    for p in glob.glob(f'{root}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'k_row_stats' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(d, k, 'dispatches', len(v), 'avg', sum(v)/len(v))
PY
find $O -name "*.csv" -size +2M -delete

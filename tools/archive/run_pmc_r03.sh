#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) of k_row_stats<512,5> in kbench: round-2 kernel (kb_base) vs round-3 (kb_new)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc; mkdir -p $O
for b in kb_base kb_new; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${b}_$i -- env KB_UNIFORM=1 $R/tools/kb/$b 512 10000 512 > $O/${b}_$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_pmc'
out=collections.OrderedDict()
for b in ('kb_base','kb_new'):
    for i in (1,2,3):
        acc=collections.defaultdict(list)
        for p in glob.glob(f'{root}/{b}_{i}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(p)):
                if 'k_row_stats' in r['Kernel_Name'] and r.get('Grid_Size','') in ('262144',''):
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            out.setdefault(k,{})[b]=(len(v), sum(v)/len(v))
with open(root+'/summary.txt','w') as f:
    for k,v in out.items():
        line=f"{k:24s} " + "  ".join(f"{b}: n={v[b][0]} avg={v[b][1]:.1f} per_wave={v[b][1]/4096:.1f}" for b in v)
        print(line); f.write(line+"\n")
PY
find $O -name "*.csv" -size +1M -delete

cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
for mode in async sync; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -- python $R/tools/overhead_trace.py $mode 10 > $O/$mode.log 2>&1
  f=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  echo "== $mode, a report every 10th step"; python $R/tools/overhead_trace.py --analyze $f
done
find $O -name "*.csv" -size +2M -delete

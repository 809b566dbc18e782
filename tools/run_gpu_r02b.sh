set -x
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs > $O/prof_stats.log 2>&1
tail -2 $O/prof_stats.log
find $O -name "*kernel_stats.csv" | head -3
cat $(find $O -name "*kernel_stats.csv" | head -1) | head -12
find $O -name "*kernel_trace.csv" -size +3M -delete

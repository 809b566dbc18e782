cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r06probe; mkdir -p $O
run() { name=$1; shift; echo "##### $name: $* $EXTRA"; env "$@" timeout 200 python tools/trace_overhead_probe.py --rounds 6 $EXTRA 2>&1 | grep -v amdgpu.ids | tee $O/$name.txt | grep -E "^ +[0-9]|traced steps"; }
run base A=1
run sigpool ROC_SIGNAL_POOL_SIZE=4096
run aqlsize ROC_AQL_QUEUE_SIZE=65536
run nointr HSA_ENABLE_INTERRUPT=0
run hostkernarg HIP_FORCE_DEV_KERNARG=0
run buffered NVRX_DEBUG_KTRACE_DELIVERY=buffer
run nocount NVRX_DEBUG_KTRACE_COUNT=0
run base_again A=1

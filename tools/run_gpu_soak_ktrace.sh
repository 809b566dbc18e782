cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_01_ktrace_datapath.py -q -s -k "soak" 2>&1 | tail -n 12 | cut -c1-700

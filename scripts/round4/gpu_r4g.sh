#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
echo "== the selection that crashed"
timeout 900 python -m pytest tests -q -s -m gpu -x -p no:cacheprovider -k "packed or c3 or smoke or gemm or graph" 2>&1 | grep -v "^  File\|^$" | tail -5 | cut -c1-300
echo "== full pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --durations=6 -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_gpu.log | cut -c1-250
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== done"

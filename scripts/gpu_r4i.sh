#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "forward_relative_key or long_sequences or head_sizes or packed or c3 or smoke or c1 or released_shape or wide" 2>&1 | tail -3 | tee $OUT/pytest_sel.log
timeout 200 python scripts/debug_img.py released 2>&1 | grep -E "==|ctx|v  |eps"
echo "== done"

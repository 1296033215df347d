#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "arbitrary_masks or head_sizes or huggingface or smoke or masked_tail or batch" 2>&1 | tail -15 | cut -c1-250 | tee $OUT/pytest_sel.log
echo "== done"

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/stamps_big.py 2>&1 | grep -v amdgpu.ids | tee $OUT/big_stamps.log | head -${HEAD:-130}
echo "== done"

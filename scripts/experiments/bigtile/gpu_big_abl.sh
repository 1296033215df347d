#!/bin/bash
# ablation builds of gemm_big.hip (python -m foldingdiff_amd.build bd<N> FDMI_BIG_DBG=<N>): per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/big_abl.log
for v in ${VARIANTS:-. bd1 bd2 bd4 bd8 bd3}; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib FDMI_GEMM_BIG=${MASK:-17} STEPS=3 TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/big_abl.log
done
echo "== done"

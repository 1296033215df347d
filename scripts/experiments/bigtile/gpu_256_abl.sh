#!/bin/bash
# ablation builds of gemm_img256.hip (python -m foldingdiff_amd.build d<N> FDMI_256_DBG=<N>): per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/g256_abl.log
for v in ${VARIANTS:-. d1 d4 d5}; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib FDMI_GEMM_256=${MASK:-17} STEPS=3 TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/g256_abl.log
done
echo "== done"

#!/bin/bash
# variants of the ping-pong GEMM, all with FDMI_GEMM_PP=$MASK:  VARIANTS=". pp3 nl4" bash scripts/gpu_ppv.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
L=$OUT/ppv.log; : > $L
FDMI_GEMM_PP=0 TAG="lockstep" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $L
for r in $(seq 1 ${ROUNDS:-1}); do
  for v in ${VARIANTS:-.}; do
    FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so FDMI_GEMM_PP=${MASK:-63} TAG="pp:$v" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $L
  done
done
for v in ${CHECK_VARIANTS:-}; do
  FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so FDMI_GEMM_PP=${MASK:-63} timeout 120 python scripts/debug_img.py released 2>&1 | grep -E "==|h_out|eps|rror" | sed "s/^/[$v] /" | tee -a $L
done
echo "== done" | tee -a $L

#!/bin/bash
# ping-pong GEMM (gemm_pp.hip) vs the lockstep kernel: stage parity, kernel times, parity tests.  MASKS="0 63" etc.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
L=$OUT/pp.log; : > $L
for cfg in ${DEBUG_CFGS:-released mini small ragged}; do
  FDMI_GEMM_PP=${CHECK_MASK:-63} timeout 120 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|  a |h_out|eps|rror|Traceback" | sed "s/^/[pp] /" | tee -a $L
done
for r in $(seq 1 ${ROUNDS:-2}); do
  for m in ${MASKS:-0 63}; do
    FDMI_GEMM_PP=$m TAG="pp=$m" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $L
  done
done
if [ -n "${PYTEST_K:-}" ]; then
  FDMI_GEMM_PP=${CHECK_MASK:-63} timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$PYTEST_K" 2>&1 | tail -8 | tee -a $L
fi
echo "== done" | tee -a $L

#!/bin/bash
# One gpurun call while bringing up the row-image path: stage-by-stage dumps + the kernel unit tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in ${DEBUG_CFGS:-small mini released ragged long}; do
  timeout 300 python scripts/debug_img.py $cfg 2>&1 | tail -16
done | tee $OUT/debug_img.log
if [ "${WITH_SAFE:-1}" = "1" ]; then
  echo "== FDMI_ATTN_SAFE=1"
  FDMI_ATTN_SAFE=1 timeout 300 python scripts/debug_img.py small released 2>&1 | grep -E "==|ctx|eps" | tee $OUT/debug_img_safe.log
fi
echo "== pytest"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "${PYTEST_K:-gemm_kernel_vs_fp64 or gemm_layernorm}" 2>&1 | tail -${PYTEST_TAIL:-25} | tee $OUT/pytest_sel.log
echo "== done"

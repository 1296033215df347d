#!/bin/bash
# round 5: d_model 192 K-tile race fix (barrier between slices 0 and 1 of stage 0): repeated bit-identity checks
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5n
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5n
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "fused_projection" 2>&1 | tail -3
done | tee $O/pytest_rep.log
timeout 600 python scripts/round5/sa_check.py 2>&1 | tail -3 | tee $O/sa_check.log
for v in . noedges; do
  TAG="c2 fused $v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done 2>&1 | tee $O/ab.log

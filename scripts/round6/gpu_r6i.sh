#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6m
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6m
CASES=0,1,2,3,5,7 timeout 900 python scripts/round6/sa16_check.py 2>&1 | tail -14 | tee $O/sa16_check.log
for rep in 1 2; do
for fa in 1 2; do
  TAG="c2 fuse_attn=$fa" FDMI_FUSE_ATTN=$fa timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done
done 2>&1 | tee $O/times.log
timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | tail -60 | tee $O/stamps.log

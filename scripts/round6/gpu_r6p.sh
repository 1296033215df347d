#!/bin/bash
# round 6: the GPU suite with seq_attn16 as the default fused kernel; C3 chunk shapes with / without it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6p
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6p
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $O/pytest_gpu.log
for fa in 0 1; do
  TAG="fuse_attn=$fa" FDMI_FUSE_ATTN=$fa timeout 300 python scripts/c3_times.py 2>&1 | grep "chunk\|c2:\|b8" | cut -c1-330
done 2>&1 | tee $O/c3_times.log

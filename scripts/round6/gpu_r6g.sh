#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6g
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6g
timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | grep -A12 "^wave 0\|^wave 4" | grep -v "slot 1[0-9]\|slot 2[0-9]" | tee $O/stamps.log

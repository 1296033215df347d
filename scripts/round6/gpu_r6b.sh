#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6b
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6b
timeout 300 python scripts/round6/sa16_stamps.py 2>&1 | tail -90 | tee $O/stamps.log

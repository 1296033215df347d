#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6q
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r6q/pytest_gpu.log

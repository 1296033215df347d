#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5f
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
FDMI_LIB=$PWD/foldingdiff_amd/_lib/sa_dump/libfdmi.so timeout 300 python scripts/round5/sa_dump.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5f/dump.log
cat gpurun_out/r5f/dump.log

#!/bin/bash
# round 4: the weight-stationary GEMM as BertIntermediate.dense inside the step (GELU epilogue), per-kernel times at C2 / C3 / C5
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4k
L=$PWD/foldingdiff_amd/_lib
{
TAG=tile FDMI_GEMM_WS=0 timeout 200 python scripts/c3_times.py
TAG=ws FDMI_GEMM_WS=1 timeout 200 python scripts/c3_times.py
TAG=ws_aux2 FDMI_GEMM_WS=1 FDMI_LIB=$L/aux2/libfdmi.so timeout 200 python scripts/c3_times.py
TAG=ws_late0aux2 FDMI_GEMM_WS=1 FDMI_LIB=$L/late0aux2/libfdmi.so timeout 200 python scripts/c3_times.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4k/times.log
cat gpurun_out/r4k/times.log

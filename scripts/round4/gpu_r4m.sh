#!/bin/bash
# round 4: q | k | v on the weight-stationary kernel -- bit identity at model level, the GPU suite, small-batch step times
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4m
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "few_rows or ws_bit or head_sizes or arbitrary_masks" > gpurun_out/r4m/pytest_ws.log 2>&1; echo "rc=$?" >> gpurun_out/r4m/pytest_ws.log
tail -5 gpurun_out/r4m/pytest_ws.log
{
TAG=auto timeout 200 python scripts/small_batch_times.py
TAG=tile FDMI_GEMM_WS=0 timeout 200 python scripts/small_batch_times.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4m/small_batch.log
cat gpurun_out/r4m/small_batch.log

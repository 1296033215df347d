#!/bin/bash
# round 4: the weight-stationary GEMM as the few-rows path -- full GPU suite, then small-batch step times with and without it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4l/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4l/pytest_gpu.log
tail -4 gpurun_out/r4l/pytest_gpu.log
{
TAG=auto timeout 200 python scripts/small_batch_times.py
TAG=tile FDMI_GEMM_WS=0 timeout 200 python scripts/small_batch_times.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l/small_batch.log
cat gpurun_out/r4l/small_batch.log

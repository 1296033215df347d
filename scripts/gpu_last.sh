#!/bin/bash
# One short call: stage parity of the default library first, then same-box A/B of library variants, then more parity.
#   VARIANTS=". touch2 notouch" bash scripts/gpu_last.sh        (everything is appended to gpurun_out/last.log as it arrives)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
L=$OUT/last.log; : > $L
dbg() { FDMI_LIB=$PWD/foldingdiff_amd/_lib/$1/libfdmi.so timeout 120 python scripts/debug_img.py $2 2>&1 | grep -E "==|  a |h_out|eps" | sed "s/^/[$1] /" | tee -a $L; }
kt() { FDMI_LIB=$PWD/foldingdiff_amd/_lib/$1/libfdmi.so TAG="$1" timeout 120 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $L; }
dbg . released; dbg . mini
for v in ${VARIANTS:-.}; do kt $v; done
dbg . ragged; dbg . small
for v in ${VARIANTS:-.}; do kt $v; done
for v in ${CHECK_VARIANTS:-}; do dbg $v released; dbg $v mini; done
timeout 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "${PYTEST_K:-gemm_layernorm or forward_released or forward_shapes or steps_c3_chunk or varlen}" 2>&1 | tail -4 | tee -a $L
echo "== done" | tee -a $L

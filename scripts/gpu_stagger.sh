export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/stagger.log
for r in 1 2; do for st in 0 4000 8000 12000; do FDMI_STAGGER=$st TAG="stagger=$st" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a gpurun_out/stagger.log; done; done
FDMI_STAGGER=8000 timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_stagger.log
grep -A16 "GEMM epilogue 2" gpurun_out/stamps_stagger.log | sed -n 3,18p

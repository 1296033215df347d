export TMPDIR=/tmp
for v in dbg1 dbg2 dbg3 dbg4; do
  echo "#### $v"
  FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so NPOS=40 python scripts/stamps_pp.py 2>&1 | grep -A40 '== GELU' | awk 'NR<=1 || (NR>=3 && NR<=8) || (NR>=15 && NR<=19) || (NR>=27 && NR<=31)'
done > gpurun_out/pp_dbg.log 2>&1

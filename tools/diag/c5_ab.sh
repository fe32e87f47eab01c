#!/bin/bash
# One GPU's share of BASELINE configs[4] (B=512, T=256, V=8000, S<=64), f32 and bf16, with the compact lattice's probability ring at
# four blocks (default where it lets a second workgroup onto the CU) and at eight (E2E_F1_RING=8); the compact lattice alone
# (B=512, T=256, V=65); a narrow alphabet whose ring fits either way; kernel statistics of both dtypes under rocprofv3.
cd "$(dirname "$0")/../.."; ROOT=$PWD
for dt in f32 bf16; do python tools/diag/wide_time.py $dt 2>&1 | tail -1; echo -n "  ring of 8: "; E2E_F1_RING=8 python tools/diag/wide_time.py $dt 2>&1 | tail -1; done
python tools/diag/time_shape.py 512 256 65 64 20 | tail -1; echo -n "  ring of 8: "; E2E_F1_RING=8 python tools/diag/time_shape.py 512 256 65 64 20 | tail -1
python tools/diag/time_shape.py 1024 256 29 64 20 | tail -1; echo -n "  ring of 4: "; E2E_F1_RING=4 python tools/diag/time_shape.py 1024 256 29 64 20 | tail -1
if [ -n "$1" ]; then
  cd /tmp; export TMPDIR=/tmp
  for dt in f32 bf16; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$1_c5_$dt -o c5 -- python $ROOT/tools/diag/profile_c5.py $dt > /dev/null 2>&1
    cp $(find $ROOT/gpurun_out/$1_c5_$dt -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/$1_c5_kernel_stats_$dt.csv
  done
fi

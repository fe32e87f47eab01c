#!/bin/bash
# builds a variant of the library with extra -D options next to the repo root: tools/diag/build_variant.sh NAME "-DFOO -DBAR"
set -e
cd "$(dirname "$0")/../../end2end_amd/csrc"
mkdir -p /tmp/e2e_var_$1
for f in *.hip; do extra=; [ $f = ctc_loss_fast.hip ] && extra=-fno-slp-vectorize; /opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -ffp-contract=off -c $f -o /tmp/e2e_var_$1/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out_ab_$1.so /tmp/e2e_var_$1/*.o -lz

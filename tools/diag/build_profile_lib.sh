#!/bin/bash
# builds the instrumented variant of the library (-DE2E_FAST_PROFILE -DE2E_BEAM_PROFILE) next to the repo root as build/diag/prof_lib.so; E2E_EXTRA_DEFS adds -D options (e.g. -DE2E_ZTOL=1e9 switches the segment self-check off)
set -e
cd "$(dirname "$0")/../../end2end_amd/csrc"; mkdir -p ../../build/diag
mkdir -p /tmp/e2e_prof
for f in *.hip; do extra=; [ $f = ctc_loss_fast.hip ] && extra=-fno-slp-vectorize; /opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DE2E_FAST_PROFILE -DE2E_BEAM_PROFILE $E2E_EXTRA_DEFS -ffp-contract=off -c $f -o /tmp/e2e_prof/${f%.hip}.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/diag/prof_lib.so /tmp/e2e_prof/*.o -lz

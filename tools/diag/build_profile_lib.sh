#!/bin/bash
# builds the instrumented variant of the library (-DE2E_FAST_PROFILE -DE2E_BEAM_PROFILE) as build/diag/prof_lib.so (or
# build/diag/$PROF_OUT); E2E_EXTRA_DEFS adds -D options (e.g. -DE2E_ZTOL=1e9 switches the segment self-check off);
# PROF_FILES limits the sources compiled with the options to the listed ones (the others come from the regular build)
set -e
cd "$(dirname "$0")/../../end2end_amd/csrc"; mkdir -p ../../build/diag
out=${PROF_OUT:-prof_lib.so}; tmp=/tmp/e2e_prof_${out%.so}; mkdir -p $tmp
objs=
for f in *.hip; do
  if [ -z "$PROF_FILES" ] || [[ " $PROF_FILES " == *" $f "* ]]; then
    ( extra=; [[ $f == ctc_loss_fast*.hip ]] && extra=-fno-slp-vectorize
      /opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DE2E_FAST_PROFILE -DE2E_BEAM_PROFILE $E2E_EXTRA_DEFS -ffp-contract=off -c $f -o $tmp/${f%.hip}.o ) &
    objs="$objs $tmp/${f%.hip}.o"
  else objs="$objs ${f%.hip}.o"; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/diag/$out $objs -lz

#!/bin/bash
# A variant of the library for same-process A/B timing (tools/diag/ab_time.py):
#   tools/diag/build_variant.sh NAME "-DFOO -DBAR" [file.hip ...]
# recompiles the listed sources (default: ctc_loss_fast.hip) with the extra options, links them with the objects of the
# regular build (run `make -C end2end_amd/csrc` first) and leaves build/diag/ab_NAME.so (build/ is git-ignored, but
# travels to the GPU box).
set -e
cd "$(dirname "$0")/../../end2end_amd/csrc"; mkdir -p ../../build/diag /tmp/e2e_var_$1
name=$1; defs=$2; shift; shift || true
files=${@:-ctc_loss_fast.hip}
objs=
for f in *.hip; do
  if [[ " $files " == *" $f "* ]]; then
    extra=; [[ $f == ctc_loss_fast*.hip ]] && extra=-fno-slp-vectorize
    /opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 $defs -ffp-contract=off -c $f -o /tmp/e2e_var_$name/${f%.hip}.o &
    objs="$objs /tmp/e2e_var_$name/${f%.hip}.o"
  else objs="$objs ${f%.hip}.o"; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/diag/ab_$name.so $objs -lz

#!/bin/bash
# profiles/ubench/build_b2dbg.sh - the library with the sliced bond kernels' in-kernel stamps compiled in (-DMPST_B2_DEBUG: mpst_debug_b2),
# beside the product library, for profiles/ubench/b2_stamps.py.  Not shipped, not loaded by the package.
set -euo pipefail
here=$(cd $(dirname $0) && pwd); src=$here/../../mpstime.jl_amd/csrc; out=$here/b2dbg${1:+_$1}; extra=${2:-}
mkdir -p $out
make -C $src -j8 > /dev/null
for f in mpst_fused mpst_api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMPST_B2_DEBUG $extra -c $src/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $out/mpst_fused.o $out/mpst_api.o $src/mpst_kernels.o $src/mpst_eig.o $src/mpst_eig_blocked.o \
  $src/mpst_encode.o $src/mpst_allreduce.o $src/mpst_impute.o $src/mpst_typed.o -o $out/libmpstime_hip_b2dbg.so -ldl
ls -la $out/libmpstime_hip_b2dbg.so

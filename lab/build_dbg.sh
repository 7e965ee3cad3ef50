#!/bin/bash
# builds a bring-up variant of the library into lab/ with extra -D flags: lab/build_dbg.sh b2dbg -DMPST_B2_DEBUG
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/mpst_build_$name
mkdir -p $out
cd $root/mpstime.jl_amd/csrc
for f in mpst_kernels mpst_fused mpst_eig mpst_eig_blocked mpst_encode mpst_allreduce mpst_impute mpst_api; do
  extra=""
  if [ $f = mpst_impute ] || [ $f = mpst_kernels ]; then extra="-mllvm -amdgpu-mfma-vgpr-form"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" $extra -c $f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $out/*.o -o $root/lab/libmpstime_hip_$name.so -L/opt/rocm/lib -lrocsolver -lrocblas -ldl -Wl,-rpath,/opt/rocm/lib
ls -la $root/lab/libmpstime_hip_$name.so

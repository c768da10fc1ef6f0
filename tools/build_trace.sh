#!/bin/bash
# development build of the library with per-phase cycle counters in the Gram kernel (tools/gram_trace.py)
set -e
cd "$(dirname "$0")/../laplace_amd/csrc"
mkdir -p build_trace
for f in lk_diag lk_eigh lk_gram lk_kron lk_lik lk_ll lk_vjp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLK_GRAM_TRACE -I../../include -I. -Wno-unused-function -c $f.hip -o build_trace/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -x hip -c lk_error.cpp -o build_trace/lk_error.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_trace/*.o -o liblaplace_hip_trace.so

#!/bin/bash
# diagnostic build of the library with the GEMM ablation switches compiled in (-DMAED_GEMM_ABLATE) -> maed_amd/libmaed_hip_ablate.so
# (selected at run time with MAED_HIP_LIB=...; scripts/gpu_gemm_ablate.sh).  Build container only (hipcc cross-compiles).
set -e
cd "$(dirname "$0")/.."
python -m maed_amd.build
cd maed_amd/csrc
for f in gemm gemm256 gemm_tn gemm_x3p; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DMAED_GEMM_ABLATE -c $f.hip -o build/${f}_ablate.o; done
objs=$(ls build/*.o | grep -v "/gemm.o\|/gemm256.o\|/gemm_tn.o\|/gemm_x3p.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmaed_hip_ablate.so $objs
ls -la ../libmaed_hip_ablate.so

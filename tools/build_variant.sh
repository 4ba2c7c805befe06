#!/bin/bash
# Build a variant of libgsplat_hip.so with extra -D flags for render.hip (kernel experiments):
#   tools/build_variant.sh NAME -DGS_BWD4_BATCH=48 -DGS_BWD4_CAP=288   ->  gaussian-splatting_amd/lib/var_NAME.so
# Run with GSPLAT_LIB=$PWD/gaussian-splatting_amd/lib/var_NAME.so python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../gaussian-splatting_amd/csrc"
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c render.hip -o /tmp/var_$name/render.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_$name.so capi.o preprocess.o scan.o binning.o radix_sort.o /tmp/var_$name/render.o loss.o optimizer.o ply.o knn.o
echo built ../lib/var_$name.so

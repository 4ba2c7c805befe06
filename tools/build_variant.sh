#!/bin/bash
# Build a variant of libgsplat_hip.so with extra -D flags for ONE kernel source (kernel experiments):
#   tools/build_variant.sh NAME -DGS_BWD_POOL=448                      (render.hip)
#   VARIANT_SRC=preprocess tools/build_variant.sh NAME -DGS_PF_OCC=5   (preprocess.hip; built without SLP packing like the Makefile does)
#   ->  gaussian-splatting_amd/lib/var_NAME.so ; run with GSPLAT_LIB=$PWD/gaussian-splatting_amd/lib/var_NAME.so python bench.py ...
set -e
name=$1; shift
src=${VARIANT_SRC:-render}
cd "$(dirname "$0")/../gaussian-splatting_amd/csrc"
mkdir -p /tmp/var_$name
extra=""
if [ "$src" = preprocess ] || [ "$src" = loss ]; then extra="-fno-slp-vectorize"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $extra "$@" -c $src.hip -o /tmp/var_$name/$src.o
objs=""
for o in capi preprocess scan binning radix_sort render loss optimizer ply knn; do
  if [ "$o" = "$src" ]; then objs="$objs /tmp/var_$name/$src.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_$name.so $objs
echo built ../lib/var_$name.so

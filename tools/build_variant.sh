#!/bin/bash
# Build a tuning variant of the library: bash tools/build_variant.sh <name> "<extra hipcc flags>"
# -> variants/lib_<name>.so (git-ignored; select with SW_LIB_PATH=variants/lib_<name>.so)
set -e
cd $(dirname $0)/..
N=$1; X=$2
mkdir -p variants/obj_$N
for f in sw_lstm sw_decoder sw_social sw_disc sw_wgrad sw_misc sw_modules sw_generic sw_wide sw_comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-gpu-rdc -fno-slp-vectorize $X -c socialways_amd/csrc/$f.hip -o variants/obj_$N/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/obj_$N/*.o -o variants/lib_$N.so
rm -rf variants/obj_$N
ls -la variants/lib_$N.so

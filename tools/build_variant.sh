#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." — development build of the library with extra flags: zlib.es_amd/libzes_NAME.so
# (use with ZES_LIB=zlib.es_amd/libzes_NAME.so; *.so is git-ignored but travels with gpurun)
set -e
cd "$(dirname "$0")/../zlib.es_amd/csrc"
name=$1; shift
mkdir -p /tmp/zesv_$name
for f in zes_api zes_deflate zes_index zes_inflate zes_inflate_par; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function $@ -c $f.hip -o /tmp/zesv_$name/$f.o &
done
wait
gcc -O2 -fPIC -c zes_gen.c -o /tmp/zesv_$name/zes_gen.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzes_$name.so /tmp/zesv_$name/*.o
echo built zlib.es_amd/libzes_$name.so

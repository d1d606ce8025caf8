#!/bin/bash
# one variant of ONE csrc file linked with the shipped objects: tools/build_file_variant.sh <tag> <file.hip> <-D flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; f=$2; shift; shift
o=$(basename $f .hip).o
mkdir -p summerset_amd/variants/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c summerset_amd/csrc/$f -o summerset_amd/variants/$tag/$o
objs=$(ls summerset_amd/csrc/*.o | grep -v "/$o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o summerset_amd/variants/libsummerset_hip_$tag.so $objs summerset_amd/variants/$tag/$o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo summerset_amd/variants/libsummerset_hip_$tag.so

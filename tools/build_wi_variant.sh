#!/bin/bash
# one variant of csrc/wire_ingest.hip linked with the shipped objects: tools/build_wi_variant.sh <tag> <-D flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p summerset_amd/variants/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c summerset_amd/csrc/wire_ingest.hip -o summerset_amd/variants/$tag/wire_ingest.o
objs=$(ls summerset_amd/csrc/*.o | grep -v wire_ingest.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o summerset_amd/variants/libsummerset_hip_$tag.so $objs summerset_amd/variants/$tag/wire_ingest.o
echo summerset_amd/variants/libsummerset_hip_$tag.so

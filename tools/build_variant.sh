#!/bin/bash
# Experimental build of libpof_hip with extra -D flags for ONE source: tools/build_variant.sh <tag> <source.hip> <-D...>
# -> planar_optical_flow_amd/lib/libpof_hip_<tag>.so (select it with POF_LIB_PATH).  The other objects are the regular build's.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
python -c "import __graft_entry__ as g; g.build()" > /dev/null
obj=planar_optical_flow_amd/build/$(basename ${src%.hip})_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I include -I planar_optical_flow_amd/csrc "$@" -c $src -o $obj
others=$(ls planar_optical_flow_amd/build/*.o | grep -v "_[a-z0-9]*\.o$" | grep -v "/$(basename ${src%.hip})\.o$" || true)
others=$(for f in planar_optical_flow_amd/csrc/*.hip; do b=$(basename ${f%.hip}); [ "$b" != "$(basename ${src%.hip})" ] && echo planar_optical_flow_amd/build/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o planar_optical_flow_amd/lib/libpof_hip_$tag.so $obj $others
echo built planar_optical_flow_amd/lib/libpof_hip_$tag.so

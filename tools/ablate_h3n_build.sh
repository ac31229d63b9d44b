#!/bin/bash
# Build ablated variants of the n-split kernel only (other objects are reused from the main build).
# usage: tools/ablate_h3n_build.sh NAME:DEFINE[,DEFINE..] ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  D=""; for d in ${defs//,/ }; do D="$D -D$d"; done
  mkdir -p build/obj_$name
  timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-comment \
    $D -x hip -c diner_amd/csrc/mlp_h3n.hip -o build/obj_$name/mlp_h3n.hip.o || exit 1
  objs=$(ls build/obj/*.o | grep -v mlp_h3n)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diner_amd/libdiner_hip_$name.so $objs build/obj_$name/mlp_h3n.hip.o || exit 1
  echo built $name
done

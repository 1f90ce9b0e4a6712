#!/bin/bash
# incremental build of libdopf_hip.so: one object per .hip file under build/obj, only what changed is recompiled (parallel)
# usage: scripts/fastbuild.sh [extra hipcc flags for kernels_agents.hip]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/decentralopf.jl_amd/csrc
OBJ=$ROOT/build/obj
mkdir -p $OBJ
pids=""
for f in dopf_api dopf_comm dopf_central kernels_central kernels_agents kernels_consensus; do
  src=$CSRC/$f.hip; obj=$OBJ/$f.o
  newest=$(ls -t $CSRC/*.h $ROOT/include/dopf.h $src | head -1)
  if [ ! -f $obj ] || [ $newest -nt $obj ] || { [ $f = kernels_agents ] && [ -n "$1" ]; }; then
    extra=""; [ $f = kernels_agents ] && extra="$@"
    (cd $CSRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-value -Wno-unused-result $extra $f.hip -o $obj) &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CSRC/libdopf_hip.so $OBJ/dopf_api.o $OBJ/dopf_comm.o $OBJ/dopf_central.o $OBJ/kernels_central.o $OBJ/kernels_agents.o $OBJ/kernels_consensus.o -ldl -lpthread
ls -la $CSRC/libdopf_hip.so

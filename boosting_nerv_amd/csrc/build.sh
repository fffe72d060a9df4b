#!/usr/bin/env bash
# Build libbnerv_hip.so (gfx950 only) in-tree.  Usage: build.sh [-j N]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libbnerv_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p _obj
pids=()
for f in abi conv conv4 convs convbf wgrad wgrad1 eltwise head3 stem gemm dense optim loss dwconv cem lnorm; do
  stale=0
  for h in *.h ../../include/bnerv.h; do [ $h -nt _obj/$f.o ] && stale=1; done
  if [ ! -f _obj/$f.o ] || [ $f.hip -nt _obj/$f.o ] || [ $stale = 1 ]; then
    $HIPCC $FLAGS -c $f.hip -o _obj/$f.o &
    pids+=($!)
  fi
done
if [ ! -f _obj/ans.o ] || [ ans.cpp -nt _obj/ans.o ] || [ common.h -nt _obj/ans.o ] || [ ../../include/bnerv.h -nt _obj/ans.o ]; then
  $HIPCC -x hip $FLAGS -c ans.cpp -o _obj/ans.o &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/abi.o _obj/conv.o _obj/conv4.o _obj/convs.o _obj/convbf.o _obj/wgrad.o _obj/wgrad1.o _obj/eltwise.o _obj/head3.o _obj/stem.o _obj/gemm.o _obj/dense.o _obj/optim.o _obj/loss.o _obj/dwconv.o _obj/cem.o _obj/lnorm.o _obj/ans.o
echo "built $(realpath $OUT)"

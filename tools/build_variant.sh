#!/bin/bash
# Build a variant of the library beside the product build: tools/build_variant.sh <name> ["extra hipcc flags"] [source dir]
#   -> build_ab/libscasr_<name>.so  (objects in /tmp/scasr_build_<name>; the product's objects are not touched)
# e.g. tools/build_variant.sh phase "-DSC_PHASE_DBG -DSC_PHASE_MIN_GRID=200";  a worktree of another commit as source dir
set -e
NAME=$1; EXTRA=$2; ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=${3:-$ROOT/speechcatcher_amd/csrc}
OBJ=/tmp/scasr_build_$NAME; mkdir -p $OBJ $ROOT/build_ab
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-variable $EXTRA"
pids=()
for f in gemm encoder search decoder_panel decoder_layer decoder_stream conformer streams; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $ROOT/build_ab/libscasr_$NAME.so
echo "built build_ab/libscasr_$NAME.so"

#!/bin/bash
# dev tool (CPU container): a VARIANT of the library for same-box A/B runs -- only the fused DDPG kernel objects are
# rebuilt with the extra flags, the rest is linked from the product's build directory.
# usage: bash tools/variant.sh <name> "<extra hipcc flags>" [more sources...]   -> tools/bin/lib_<name>.so
set -e
NAME=$1; EXTRA=$2; shift 2
SRCS="kernels_ddpg.hip kernels_ddpg_pair.hip $*"
cd "$(dirname "$0")/../ddrl_amd/csrc"
B=build_v_$NAME
mkdir -p $B ../../tools/bin
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-result -Wno-pass-failed -mllvm -pragma-unroll-threshold=1048576 -DDDRL_NWAVES=4 $EXTRA"
pids=""
for s in $SRCS; do /opt/rocm/bin/hipcc $FLAGS -c -x hip $s -o $B/$s.o & pids="$pids $!"; done
for p in $pids; do wait $p; done
OBJ=""
for o in ${BASE:-build}/*.o; do b=$(basename $o); if [ -f $B/$b ]; then OBJ="$OBJ $B/$b"; else OBJ="$OBJ $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/lib_$NAME.so $OBJ -ldl
ls -la ../../tools/bin/lib_$NAME.so

#!/bin/bash
# usage: bash tools/r3_stamps.sh "<variants>" "<G pair nupd>|..."
set -u
O=gpurun_out/r03/stamps; mkdir -p $O
IFS='|' read -ra CFGS <<< "$2"
for v in $1; do
  so=$PWD/tools/bin/lib_$v.so; [ "$v" = "tree" ] && so=$PWD/ddrl_amd/libddrl_pop.so
  for cfg in "${CFGS[@]}"; do
    set -- $cfg
    echo "== $v $cfg"
    DDRL_POP_SO=$so timeout -k 10 200 python3 tools/stamps2.py $1 $2 $3 2>&1 | tee $O/${v}_$1_$2_$3.txt || exit 1
  done
done
echo STAMPS_OK

#!/bin/bash
# dev tool (GPU box): per-phase stamps of the fused update at several population sizes / experiment builds
OUT=gpurun_out/r02
mkdir -p $OUT
for G in 128 256 1024; do
  for lib in "" nomem nowload nooptmem; do
    if [ -n "$lib" ]; then export DDRL_POP_SO=$PWD/tools/bin/lib$lib.so; else unset DDRL_POP_SO; fi
    echo "=== G=$G lib=${lib:-product}" >> $OUT/occupancy.log
    timeout -k 10 120 python3 tools/stamps.py $G ewc >> $OUT/occupancy.log 2>&1 || exit 1
  done
done

#!/bin/bash
# round-3 run 1: baseline stamps, membench modes, A/B of the first variants
set -u
O=gpurun_out/r03/run1; mkdir -p $O
for m in 0 2 3 4 5; do timeout -k 5 60 tools/bin/membench $m 2048 20 >> $O/membench.txt 2>&1 || exit 1; done
cat $O/membench.txt
export DDRL_POP_SO=$PWD/tools/bin/lib_base.so
timeout -k 10 200 python3 tools/stamps2.py 1024 1 8 > $O/stamps_1024_pair.txt 2>&1 || { tail -5 $O/stamps_1024_pair.txt; exit 1; }
timeout -k 10 200 python3 tools/stamps2.py 1024 0 8 > $O/stamps_1024_single.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/stamps2.py 128 0 8 > $O/stamps_128_single.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/stamps2.py 1024 0 1 > $O/stamps_1024_single_n1.txt 2>&1 || exit 1
cat $O/stamps_*.txt
unset DDRL_POP_SO
bash tools/r3_ab.sh r03/run1/ab "base mskip sets2 prio" "--steps 96 --warmup 16|--steps 96 --warmup 16 --pair 0|--steps 96 --warmup 16 --agents 128" || exit 1
echo RUN1_OK

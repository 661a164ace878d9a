#!/bin/bash
set -u
for V in "$@"; do
O=gpurun_out/r03/$V; mkdir -p $O
DDRL_POP_SO=$PWD/tools/bin/lib_$V.so timeout -k 10 600 python3 -u -m pytest tests/test_ddpg_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
done
bash tools/r3_ab.sh r03/ab_$1 "g1 $*" "--steps 96 --warmup 16|--steps 96 --warmup 16 --pair 0|--steps 96 --warmup 16 --agents 128|--steps 96 --warmup 16 --size small" || exit 1
echo RUN5_OK

#!/bin/bash
set -u
V=${1:-g1}
O=gpurun_out/r03/$V; mkdir -p $O
DDRL_POP_SO=$PWD/tools/bin/lib_$V.so timeout -k 10 600 python3 -u -m pytest tests/test_ddpg_gpu.py tests/test_agents_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
bash tools/r3_ab.sh r03/$V/ab "base $V" "--steps 96 --warmup 16|--steps 96 --warmup 16 --pair 0|--steps 96 --warmup 16 --agents 128|--steps 96 --warmup 16 --size small" || exit 1
DDRL_POP_SO=$PWD/tools/bin/lib_$V.so timeout -k 10 200 python3 tools/stamps2.py 1024 1 8 > $O/stamps_1024_pair.txt 2>&1 || exit 1
cat $O/stamps_1024_pair.txt
echo RUN3_OK

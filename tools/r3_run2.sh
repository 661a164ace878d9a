#!/bin/bash
set -u
O=gpurun_out/r03/run2; mkdir -p $O
for v in nomem nowload nooptmem; do
  for cfg in "128 0 8" "1024 0 8" "1024 1 8"; do
    set -- $cfg
    DDRL_POP_SO=$PWD/tools/bin/lib_$v.so timeout -k 10 200 python3 tools/stamps2.py $1 $2 $3 > $O/stamps_${v}_$1_$2.txt 2>&1 || { tail -5 $O/stamps_${v}_$1_$2.txt; exit 1; }
    echo "== $v"; cat $O/stamps_${v}_$1_$2.txt
  done
done
for g in 256 512; do DDRL_POP_SO=$PWD/tools/bin/lib_base.so timeout -k 10 200 python3 tools/stamps2.py $g 0 8 > $O/stamps_base_$g.txt 2>&1 || exit 1; cat $O/stamps_base_$g.txt; done
echo RUN2_OK

#!/bin/bash
# usage: bash tools/r3_pmc.sh <variant> <tag> "<counters>" ["bench args"]
set -u
V=$1; TAG=$2; CTR=$3; ARGS=${4:---steps 32 --warmup 8}
so=$GRAFT_REPO_ROOT/tools/bin/lib_$V.so; [ "$V" = "tree" ] && so=$GRAFT_REPO_ROOT/ddrl_amd/libddrl_pop.so
export DDRL_POP_SO=$so
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/pmc_$V/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTR --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu --no-extra > $OUT/log.txt 2>&1 || { echo "pass failed"; tail -5 $OUT/log.txt; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'ddpg_update' in k:
        print(k[:70]); [print('   %-34s %.5g'%(c,x)) for c,x in sorted(v.items())]
"

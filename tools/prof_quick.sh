#!/bin/bash
# dev tool (GPU box): kernel-trace stats of a short bench run;  usage: bash tools/prof_quick.sh <tag> "<bench args>"
TAG=$1; ARGS=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu --no-extra > $OUT/bench.json 2> $OUT/bench.err || exit 1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -5 $OUT/kernel_stats.csv | cut -c1-200

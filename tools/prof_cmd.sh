#!/bin/bash
# dev tool (GPU box): kernel-trace stats of any python tool;  usage: bash tools/prof_cmd.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/$TAG
mkdir -p $OUT
S=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $S "$@" > $OUT/out.json 2> $OUT/err.log || exit 1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-150

#!/bin/bash
# rocprofv3 kernel stats of the default bench workload + the bench line itself (GPU box, repo root)
# usage: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 8 --no-cpu --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || { echo "rocprof run failed"; tail -5 $OUT/rocprof.err; exit 1; }
echo STATS_OK

#!/bin/bash
# the round's closing measurements on one box (GPU box, repo root): kernel stats, bench lines, PMC passes
# usage: bash tools/final_round.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:-r02/z}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
b() { local name=$1; shift; timeout -k 10 300 python3 $R/bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "bench $name failed"; tail -3 $OUT/$name.err; exit 1; }; }
bash $R/tools/profile_round.sh $TAG || exit 1
cd $R
b bench && b bench_128 --agents 128 --no-cpu && b bench_small --size small --no-cpu && b bench_large --size large --steps 40 --warmup 4 --no-cpu && \
b bench_upl1 --updates-per-launch 1 --no-cpu && b nfac_bench --workload nfac || exit 1
bash $R/tools/prof_cmd.sh ../$TAG/nfac_prof bench.py --workload nfac --no-cpu > $OUT/nfac_prof.txt 2>&1 || { echo "nfac profile failed"; exit 1; }
cd $R
timeout -k 10 200 python3 tools/td3_bench.py > $OUT/td3_bench.txt 2>&1 || { echo "td3 bench failed"; exit 1; }
timeout -k 10 200 python3 tools/bench_bn.py > $OUT/bn_bench.json 2> $OUT/bn_bench.err || { echo "bn bench failed"; exit 1; }
bash $R/tools/pmc_passes.sh $TAG/pmc "--steps 32 --warmup 8" || exit 1
cd $R
python3 tools/pmc_summary.py gpurun_out/$TAG/pmc > $OUT/pmc_summary.json || exit 1
echo FINAL_ROUND_OK

#!/bin/bash
# timing, then one FETCH_SIZE and one WRITE_SIZE pass per mode (single launch after 3 warmups each)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/membench
mkdir -p $OUT
B=$GRAFT_REPO_ROOT/tools/bin/membench
for m in 0 1 2; do timeout -k 5 60 $B $m 2048 20 || exit 1; done
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$m -o f$m -- $B $m 2048 1 > $OUT/f$m.log 2>&1 || exit 1
  timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w$m -o w$m -- $B $m 2048 1 > $OUT/w$m.log 2>&1 || exit 1
done
echo PASSES_OK

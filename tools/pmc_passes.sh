#!/bin/bash
# PMC passes over the bench workload (one counter group per run, as MI355X_MICROARCH.md prescribes).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> ["<bench.py arguments>"]
set -u
TAG=${1:-pmc}
ARGS=${2:---steps 32 --warmup 8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu --no-extra > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 $OUT/$1.log; exit 1; }
}
run fetch "FETCH_SIZE" && run write "WRITE_SIZE" && \
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" && \
run sq2 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VMEM" && \
echo ALL_PASSES_OK

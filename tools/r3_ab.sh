#!/bin/bash
# dev tool (GPU box, repo root): same-box A/B of library variants built by tools/variant.sh
# usage: bash tools/r3_ab.sh <outdir under gpurun_out> "<variant names>" "<bench arg sets separated by |>"
set -u
OUT=gpurun_out/$1; VARS=$2; SETS=$3
mkdir -p $OUT
IFS='|' read -ra ARGSETS <<< "$SETS"
k=0
for args in "${ARGSETS[@]}"; do
  k=$((k+1))
  for v in $VARS; do
    so=$PWD/tools/bin/lib_$v.so
    [ "$v" = "tree" ] && so=$PWD/ddrl_amd/libddrl_pop.so
    DDRL_POP_SO=$so timeout -k 10 240 python3 bench.py --no-cpu --no-extra $args > $OUT/${v}_$k.json 2> $OUT/${v}_$k.err || { echo "bench $v [$args] failed"; tail -3 $OUT/${v}_$k.err; exit 1; }
    python3 - "$OUT/${v}_$k.json" "$v" "$args" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s %-44s %9.0f upd/s  ms/step %.4f  kernel_ms %.4f  frac %.3f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r.get("kernel_avg_ms", 0), r.get("frac", 0)), flush=True)
PY
  done
done

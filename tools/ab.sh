# same-box A/B: the tree's library against tools/bin/libbase.so (a build of another commit), bench lines side by side
set -e
mkdir -p gpurun_out/r02/ab
run() { # name, "base" or "", bench arguments
  local name=$1; shift; local so=$1; shift
  if [ -n "$so" ]; then DDRL_POP_SO=$PWD/tools/bin/libbase.so timeout -k 10 200 python bench.py --no-cpu "$@" > gpurun_out/r02/ab/$name.json 2>/dev/null
  else timeout -k 10 200 python bench.py --no-cpu "$@" > gpurun_out/r02/ab/$name.json 2>/dev/null; fi
}
run new_def ""
run base_def base
run new_def2 ""
run base_def2 base
run new_128 "" --agents 128
run base_128 base --agents 128
run new_small "" --size small
run base_small base --size small
run new_nfac "" --workload nfac
run base_nfac base --workload nfac
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r02/ab/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["roofline"].get("kernel_avg_ms"))
PY

#!/bin/bash
# A/B runs of bench.py on one box: tools/ab_bench.sh TAG ENVVAR v1 v2 [v1 v2 ...] [-- extra bench args]
# prints ms/step, pm_forces and the stage averages of every run; JSON lines under gpurun_out/ab_TAG/
tag=$1; var=$2; shift 2
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out/ab_$tag
i=0
for v in "${vals[@]}"; do
  f=gpurun_out/ab_$tag/${var}_${v}_$i.json
  env $var=$v python bench.py --no-sub-record --no-cpu-baseline "$@" > $f 2> gpurun_out/ab_$tag/err_$i.log || { echo "bench failed"; tail -5 gpurun_out/ab_$tag/err_$i.log; exit 1; }
  python - "$f" "$var=$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "ms/step", d["ms_per_step"], "pm_forces", d["force_cycle"].get("pm_forces_ms"), {k: round(v["ms_per_launch"], 4) for k, v in d["stages"].items()})
PY
  i=$((i+1))
done

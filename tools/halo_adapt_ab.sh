#!/bin/bash
# the window halo chosen on the device (default) against the static rule (MCPM_PAINT_ADAPT=0): bench lines per mesh, alternating processes
OUT=${1:-gpurun_out/ha}; mkdir -p $OUT
for n in ${MESHES:-512 256 128}; do for a in 0 1 0 1; do
  ST=50; [ $n = 512 ] && ST=20
  MCPM_PAINT_ADAPT=$a timeout -k 10 300 python3 bench.py --mesh $n --steps $ST --warmup 5 --no-sub-record --no-cpu-baseline > $OUT/b_${n}_$a.json 2> $OUT/b_${n}_$a.err || exit 1
  python3 - $OUT/b_${n}_$a.json $a <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["stages"]
print(f"mesh {d['config']['mesh']} adapt {sys.argv[2]}: {d['value']:.2f} steps/s, {d['ms_per_step']:.4f} ms/step, paint {st['paint']['ms_per_launch']:.4f}, paint3 {st['paint3']['ms_per_launch']:.4f}, pm_forces {d['force_cycle']['pm_forces_ms']}")
PY
done; done | tee $OUT/summary.txt

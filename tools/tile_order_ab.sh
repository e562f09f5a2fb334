#!/bin/bash
# A/B of the order in which an XCD walks the tile pencils of its slab (MCPM_TILE_ORDER, paint_tiled.hip::tile_of_block):
# stage times of the two tiled paints over the bench trajectory, alternating processes.   usage: tools/tile_order_ab.sh [mesh] [outdir]
N=${1:-512}; OUT=${2:-gpurun_out/to}; mkdir -p $OUT
for rep in ${REPS:-1 2}; do for o in ${ORDERS:-0 1}; do
  MCPM_TILE_ORDER=$o timeout -k 10 300 python3 bench.py --mesh $N --steps 10 --warmup 3 --no-sub-record --no-cpu-baseline > $OUT/b_${N}_${o}_$rep.json 2> $OUT/b_${N}_${o}_$rep.err || exit 1
  python3 - $OUT/b_${N}_${o}_$rep.json $o <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["stages"]
print(f"order {sys.argv[2]}: {d['value']:.2f} steps/s, {d['ms_per_step']:.3f} ms/step, paint {st['paint']['ms_per_launch']:.4f}, paint3 {st['paint3']['ms_per_launch']:.4f}, kspace {st['kspace']['ms_per_launch']:.4f}, "
      f"kick_drift {st['kick_drift']['ms_per_launch']:.4f}, step_adjoint {st['step_adjoint']['ms_per_launch']:.4f}, pm_forces {d.get('force_cycle', {}).get('pm_forces_ms')}")
PY
done; done | tee $OUT/summary_$N.txt

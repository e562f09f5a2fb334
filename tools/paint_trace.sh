#!/bin/bash
# durations of the tiled-paint kernels, call by call, over one bench trajectory (rocprofv3 --kernel-trace)
N=${1:-512}; O=${2:-gpurun_out/ptr}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o t -- python3 $R/bench.py --mesh $N --steps 10 --warmup 0 --no-sub-record --no-cpu-baseline > $R/$O/run.log 2>&1 || exit 1
python3 - $R/$O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    for key in ("paint_tile_kernel", "paint3_tile_kernel", "box_tile_kernel", "coverage_duty_kernel", "tile_prologue_kernel"):
        if key in k:
            out.setdefault(k[:60], []).append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
for k, v in out.items():
    print(k, "n=%d" % len(v), "last 24:", v[-24:])
PY
rm -rf $R/$O/tr

#!/bin/bash
# kernel-to-kernel gaps inside pm_forces: rocprofv3 --kernel-trace of tools/time_pm_forces.py, start / end timestamps of the last call's kernels
N=${1:-256}; O=${2:-gpurun_out/gaps}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o t -- python3 $R/tools/time_pm_forces.py $N 1 > $R/$O/run.log 2>&1 || exit 1
python3 - $R/$O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last read3 kernel ends a pm_forces call; print the kernels of that call
idx = [i for i, r in enumerate(rows) if "read3_il" in r["Kernel_Name"]]
end = idx[-1]; start = idx[-2] + 1
prev = int(rows[start - 1]["End_Timestamp"])
t0 = int(rows[start]["Start_Timestamp"])
tot_k = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'].split('(')[0][:44]:44s} gap before {(s - prev) / 1e3:7.1f} us   duration {(e - s) / 1e3:7.1f} us")
    tot_k += e - s
    prev = e
print(f"call: wall {(prev - t0) / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us, gaps {(prev - t0 - tot_k) / 1e3:.1f} us")
PY
rm -rf $R/$O/tr

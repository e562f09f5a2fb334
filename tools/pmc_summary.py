"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.  usage: pmc_summary.py DIR [kernel-substring ...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:44s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")

"""Aggregates a rocprofv3 --pmc counter_collection CSV by kernel name: mean counter value per dispatch."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
keep = sys.argv[2:] or ["gemm_win", "attn_flash", "rownorm"]
for k in agg:
    if not any(s in k for s in keep):
        continue
    n = max(cnt[k].values())
    if n < 3:
        continue
    print(k, "dispatches", n)
    for c in sorted(agg[k]):
        print(f"   {c:32s} {agg[k][c] / cnt[k][c]:16.1f}")

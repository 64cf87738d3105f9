"""Aggregates a rocprofv3 kernel-trace CSV by (kernel, grid): second half of the trace (steady state)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    key = (r["Kernel_Name"][:52], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k[0]:52s} grid=({k[1]},{k[2]},{k[3]}) n={v[0]:5d} avg={v[1]/v[0]:7.2f}us tot={v[1]/1e3:7.2f}ms {v[1]/tot*100:5.1f}%")
print("total ms", round(tot / 1e3, 2), "kernels", len(rows))

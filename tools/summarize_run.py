"""Prints the family / instantiation shares of a rocprofv3 kernel-stats csv and the keys of a bench.py JSON line (used to
write profiles/README.md and DESIGN.md)."""
import csv
import json
import re
import sys

d = sys.argv[1]
for name in ("bench", "longform"):
    rows = list(csv.DictReader(open(f"{d}/{name}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {name}: total GPU {tot / 1e6:.1f} ms")
    fam = {}
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        f = re.match(r"(?:void )?([A-Za-z_0-9:]+)", n).group(1)
        e = fam.setdefault(f, [0, 0.0])
        e[0] += int(r["Calls"])
        e[1] += float(r["TotalDurationNs"])
    for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:7]:
        print(f"  {100 * t / tot:5.1f}%  calls {c:7d}  avg {t / c / 1e3:7.1f} us  {f}")
    for r in rows[:6]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")[:64]
        print(f"     {float(r['Percentage']):5.1f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:7.1f} us  {n}")
try:
    j = json.loads(open(f"{d}/bench_default.log").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "roofline", "roofline_attn", "roofline_lm", "continuous_batching", "single_utterance", "parity_build", "cpu_baseline"):
        print(k, json.dumps(j.get(k))[:400])
except FileNotFoundError:
    pass

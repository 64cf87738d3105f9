"""Prints the last N dispatches of a rocprofv3 --kernel-trace CSV in launch order: duration (us), grid, kernel name."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{d:7.1f} us  grid {r["Grid_Size_X"]:>8s}x{r["Grid_Size_Y"]:>4s}x{r["Grid_Size_Z"]:>3s} wg {r["Workgroup_Size_X"]:>4s}  {r["Kernel_Name"][:90]}')

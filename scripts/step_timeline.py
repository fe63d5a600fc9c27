"""Kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV: name, duration, gap to the previous kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
for r in rows[-n:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:48].ljust(48), "dur %8.1f us" % ((en - st) / 1e3), "gap %7.1f us" % (((st - prev) / 1e3) if prev else 0))
    prev = en

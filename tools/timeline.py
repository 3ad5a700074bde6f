#!/usr/bin/env python3
"""Prints the kernel timeline of one bench step from a rocprofv3 kernel-trace CSV:
usage: timeline.py <kernel_trace.csv> [steps-from-the-end] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
idx = [i for i, r in enumerate(rows) if "join_link" in r["Kernel_Name"]]
i0 = idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    n = r["Kernel_Name"].split("(")[0][-56:]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} q{r.get('Queue_Id', '?')} {n}")

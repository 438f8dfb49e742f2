#!/usr/bin/env python3
"""Per-launch durations of the staged pipeline's round kernel from a rocprofv3 kernel trace (csv):
    python tools/wf_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

rows = [x for x in csv.DictReader(open(sys.argv[1])) if "k_wf_round" in x["Kernel_Name"]]
rows = rows[len(rows) // 2:]                     # the second of the two renders
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in rows]
gap = [(int(rows[k + 1]["Start_Timestamp"]) - int(rows[k]["End_Timestamp"])) / 1e3 for k in range(len(rows) - 1)] + [0.0]
step = max(1, len(rows) // 24)
for k in list(range(0, min(8, len(rows)))) + list(range(8, len(rows), step)):
    print("launch %5d  %8.1f us  gap %6.1f us  grid %s x %s" % (k, dur[k], gap[k], rows[k]["Grid_Size_X"], rows[k]["Grid_Size_Y"]))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print("%d launches, span %.2f ms, busy %.2f ms, gaps %.2f ms" % (len(rows), span, sum(dur) / 1e3, sum(g for g in gap if g > 0) / 1e3))

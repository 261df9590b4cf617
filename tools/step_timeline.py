"""Timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV of bench.py: start (us), duration, gap to the
previous kernel, name.  usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 10]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 10
idx = [i for i, r in enumerate(rows) if "k_tokens_mask" in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))[:78]
prev, busy = t0, 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:5.1f}  {short(r['Kernel_Name'])}")
    prev, busy = e, busy + e - s
print(f"{len(step)} kernels, span {(prev - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")

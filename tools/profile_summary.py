"""Markdown summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py.
Usage: python tools/profile_summary.py <kernel_stats.csv> <out.md> "<command that was profiled>" """
import csv
import re
import sys

src, dst, cmd = sys.argv[1:4]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# rocprofv3 --kernel-trace --stats of `{cmd}` (1x MI355X)", "",
       "Hand-written kernels (libmivae) -- average duration per dispatch:", "",
       "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
for r in rows:
    m = re.search(r"\b(k_[a-z_0-9]+)", r["Name"])
    if m:
        out.append(f"| {m.group(1)} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
out += ["", "Top 15 kernels overall:", "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
for r in rows[:15]:
    name = re.sub(r"at::native::|\(anonymous namespace\)::", "", r["Name"])[:90]
    out.append(f"| {name} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
mine = sum(float(r["TotalDurationNs"]) for r in rows if re.search(r"\bk_[a-z_0-9]+", r["Name"]))
out.append(f"\nTotal GPU kernel time {tot / 1e6:.1f} ms over the whole process; hand-written kernels {100 * mine / tot:.1f} %.")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))

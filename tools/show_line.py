"""Print the headline numbers of a bench.py JSON line (file argument or stdin)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for ln in src:
    if not ln.startswith("{"):
        continue
    l = json.loads(ln)
    print(f"{l['config']['workload'][:40]:40s} value={l['value']} ms/step={l['ms_per_step']} median={l.get('ms_per_step_median')} "
          f"mode={l['step_mode']} hbm_measured={l.get('hbm_measured_GBps')} loss={l['final_loss']}")
    for r in l.get("roofline_flow") or []:
        print(f"   {r['kernel']:26s} {r['avg_us']:9.2f} us  bound={r['bound']:10s} frac={r['frac']:.4f} traffic_x={r.get('wasted_traffic')}")
    if l.get("roofline"):
        r = l["roofline"]
        print(f"   dominant {r['kernel']:17s} {r['avg_us']:9.2f} us  bound={r['bound']:10s} frac={r['frac']:.4f}")
    for name, o in (l.get("other_configs") or {}).items():
        if "error" in o:
            print(f" {name}: ERROR {o['error']}")
            continue
        print(f" {name}: value={o['value']} ms/step={o['ms_per_step']} median={o['ms_per_step_median']} mode={o['step_mode']} loss={o['final_loss']}")
        for r in o.get("roofline_flow") or []:
            print(f"   {r['kernel']:26s} {r['avg_us']:9.2f} us  bound={r['bound']:10s} frac={r['frac']:.4f} traffic_x={r.get('wasted_traffic')}")

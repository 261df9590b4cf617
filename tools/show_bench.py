import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read())
        print(f, d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items() if "lce" in k})
    except Exception as e:
        print(f, "ERR", e)

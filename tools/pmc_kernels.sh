#!/bin/bash
# usage: tools/pmc_kernels.sh <outdir-under-gpurun_out> "<counters>" <args to bench_kernels.py ...>   (GPU box)
# one rocprofv3 --pmc pass (counters within the per-block slot limits) of the kernel micro-bench; prints per-kernel means.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; ctr="$2"; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'k_big' in k or 'k_stack' in k or 'k_dw' in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'n=', len(next(iter(d.values()))))
PY

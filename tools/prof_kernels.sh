#!/bin/bash
# usage: tools/prof_kernels.sh <outdir-under-gpurun_out> <args to bench_kernels.py ...>   (run on the GPU box)
# rocprofv3 kernel trace + stats (csv) of the kernel micro-bench; prints the top kernels by total time.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY

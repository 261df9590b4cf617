#!/bin/bash
# usage: tools/prof_script.sh <outdir-under-gpurun_out> <python script under the repo> [args]   (run on the GPU box)
# rocprofv3 kernel trace + stats (csv) of one script; prints the top kernels by total time and keeps only the summary.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
script=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python3 $GRAFT_REPO_ROOT/$script "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
mkdir -p $out
cp $(find /tmp/prof_$$ -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
rm -rf /tmp/prof_$$
grep '^{' $out.log
python3 - $out <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1] + '/kernel_stats.csv')))[:16]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY

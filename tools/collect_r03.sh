#!/bin/bash
# usage (GPU box): tools/collect_r03.sh <outdir-under-gpurun_out> ["c2 c3 c5"]
# Round-3 evidence set: FETCH_SIZE / WRITE_SIZE passes of tools/pmc_target.py for c2, c3, c5 (-> profiles/pmc_traffic.json
# through tools/pmc_traffic.py), rocprofv3 kernel stats of the graph-replayed C2 step and of the eager C4 step.
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in ${2:-c2 c3 c5}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_${cfg}_$ctr -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_target.py $cfg 3 > $out/pmc_${cfg}_$ctr.log 2>&1 || { tail -5 $out/pmc_${cfg}_$ctr.log; exit 1; }
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $cfg $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE $out/pmc_traffic.json > $out/pmc_traffic_$cfg.log 2>&1 || { tail -5 $out/pmc_traffic_$cfg.log; exit 1; }
  rm -rf $out/pmc_${cfg}_FETCH_SIZE $out/pmc_${cfg}_WRITE_SIZE
done
for cfg in c2 c4; do
  steps=200; [ $cfg = c4 ] && steps=10
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-others --no-cpu-baseline --steps $steps --warmup 5 > $out/${cfg}_step_profile.log 2>&1 || { tail -5 $out/${cfg}_step_profile.log; exit 1; }
  cp $(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1) $out/${cfg}_step_kernel_stats.csv
  rm -rf /tmp/prof_$cfg
done
echo collected $out

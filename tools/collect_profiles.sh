#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <config> <outdir-under-gpurun_out>
# rocprofv3 evidence for one config of tools/bench_kernels.py (the flow kernels alone, bench shapes):
#   kernel trace + stats, two SQ counter passes, FETCH_SIZE and WRITE_SIZE passes -- every pass its own run.
# Leaves <out>/kernel_stats.csv (rocprofv3's own stats table) and <out>/pmc.json (per-kernel counter means); the raw
# traces are deleted (the directory travels back through a 64 MiB pipe).
cfg=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 args...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d $out/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py $cfg --iters 10 > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
}
run stats --kernel-trace --stats && \
run pmc_a --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT && \
run pmc_b --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES && \
run fetch --pmc FETCH_SIZE && \
run write --pmc WRITE_SIZE || exit 1
python3 - $out <<'PY'
import csv, glob, json, os, shutil, sys, collections
out = sys.argv[1]
shutil.copy(glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)[0], out + '/kernel_stats.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('pmc_a', 'pmc_b', 'fetch', 'write'):
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if 'k_big' in name or 'k_stack' in name or 'k_dw' in name or 'k_reparam' in name:
                acc[name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: dict({c: sum(v) / len(v) for c, v in d.items()}, dispatches=len(next(iter(d.values())))) for k, d in acc.items()}
json.dump(res, open(out + '/pmc.json', 'w'), indent=1, sort_keys=True)
for p in ('stats', 'pmc_a', 'pmc_b', 'fetch', 'write'):
    shutil.rmtree(f'{out}/{p}', ignore_errors=True)
print('collected', out)
PY

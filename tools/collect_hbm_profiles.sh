#!/bin/bash
# usage (GPU box): tools/collect_hbm_profiles.sh <config> <outdir-under-gpurun_out>
# rocprofv3 evidence for the HBM-bound entry points of tools/bench_hbm_kernels.py at one BASELINE latent shape:
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE, every pass its own run (eager launches: --plain).
# Leaves <out>/hbm_kernel_stats.csv (rocprofv3's own stats table) and <out>/hbm_pmc.json (per device kernel: dispatches,
# mean FETCH_SIZE / WRITE_SIZE turned into bytes as MI355X_MICROARCH.md prescribes: KiB, reads doubled on gfx950).
cfg=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 args...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d $out/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_hbm_kernels.py $cfg --plain > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
}
run stats --kernel-trace --stats && \
run fetch --pmc FETCH_SIZE && \
run write --pmc WRITE_SIZE || exit 1
python3 - $out <<'PY'
import csv, glob, json, shutil, sys, collections
out = sys.argv[1]
shutil.copy(glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)[0], out + '/hbm_kernel_stats.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('fetch', 'write'):
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            if name.startswith('k_') and 'hbm_copy' not in name:
                acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in acc.items():
    rd = 2.0 * 1024.0 * sum(d.get('FETCH_SIZE', [0])) / max(len(d.get('FETCH_SIZE', [0])), 1)
    wr = 1024.0 * sum(d.get('WRITE_SIZE', [0])) / max(len(d.get('WRITE_SIZE', [0])), 1)
    res[k] = {'dispatches': len(next(iter(d.values()))), 'read_bytes': round(rd), 'write_bytes': round(wr), 'hbm_bytes_per_launch': round(rd + wr)}
json.dump(res, open(out + '/hbm_pmc.json', 'w'), indent=1, sort_keys=True)
for p in ('stats', 'fetch', 'write'):
    shutil.rmtree(f'{out}/{p}', ignore_errors=True)
print('collected', out)
PY

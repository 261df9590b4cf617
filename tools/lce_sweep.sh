for w in 2048 4096 8192 16384; do
MIVAE_LCE_WAVES=$w timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print($w, {k:v['avg_us'] for k,v in d['kernels'].items() if k.startswith('lce')})" || exit 1
done

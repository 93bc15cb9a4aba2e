set -e
for v in "SZ_FUSE_FORCING=2" "SZ_FUSE_FORCING=1" "SZ_OVERLAP=1" "SZ_FUSE_FORCING=0"; do
  echo "== $v"
  env $v timeout -k 10 280 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-strong-reference 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('ms_per_step_min'), d.get('kernel_ms_per_step'))
"
done

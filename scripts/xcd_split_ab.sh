for v in default 1 2 4 8; do
  if [ $v = default ]; then unset EDMP_XCD_SPLIT; else export EDMP_XCD_SPLIT=$v; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD_SPLIT=$v', round(d['value']), d['ms_per_step'])"
done

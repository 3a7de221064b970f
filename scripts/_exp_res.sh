for args in "" "--resident-group 1" "--resident-group 1 --resident-rows 48" ""; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-from-host --no-scaling-model --no-ref-compat $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$args', d['value'], d['ms_per_step'], d['match_only_ms_per_batch'])"
done

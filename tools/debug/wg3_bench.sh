# whole-bench A/B of the three-workgroups-per-CU form (default: on for >= 2304-tile grids)
for r in 1 2; do
for v in 0 -1; do
  if [ $v = 0 ]; then export CDET_HALO_WG3=0; else unset CDET_HALO_WG3; fi
  echo "== CDET_HALO_WG3=${CDET_HALO_WG3:-auto}"
  python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d.get('extra',{})
print(d['value'], d['ms_per_step'], d['roofline']['frac'], e.get('north_star_fwd',{}).get('ms'), e.get('north_star_fwd',{}).get('frac'))"
done
done

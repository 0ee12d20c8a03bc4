#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/pp_bench.txt; : > $O
for r in 1 2; do
for pp in 0 1 2; do
  echo "== CDET_CONV_PP=$pp (round $r)" >> $O
  CDET_CONV_PP=$pp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline'].get('frac'),d['roofline'].get('achieved'),'ns_fwd',d.get('north_star_fwd',{}).get('ms'),d.get('north_star_fwd',{}).get('frac'),'kernel_ms_total',d.get('kernel_ms_total'))
" >> $O
done
done
cat $O

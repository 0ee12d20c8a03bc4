#!/bin/bash
# launch time as a function of the K-loop length (40 x 40 x Cin -> 320, batch 32): fixed part a and per-step part b of both 3x3 forms
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/pp_fit.txt; : > $O
SH=""; for c in 32 64 128 256 320 480 640 960 1280; do SH="$SH --shape 40,40,$c,320,3"; done
for pp in 0 2; do
  echo "== CDET_CONV_PP=$pp" >> $O
  CDET_CONV_PP=$pp python tools/conv_tiled_bench.py --rounds 5 $SH 2>&1 | grep -E "^ *[0-9]+x" >> $O
done
python3 - >> $O <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/pp_fit.txt'):
    if l.startswith('=='): cur=l.split('=')[-1].strip(); rows[cur]=[]; continue
    m=re.match(r'\s*40x40\s+(\d+)->320\s+3x3 x1\s+[\d.]+\s+\d+\s+([\d.]+)',l)
    if m: rows[cur].append((int(m.group(1))//32*9, float(m.group(2))*1e3))
import numpy as np
for k,v in rows.items():
    n=np.array([a for a,_ in v],float); t=np.array([b for _,b in v])
    b,a=np.polyfit(n[2:],t[2:],1)
    print(f"PP={k}: launch us = {a:.1f} + {b:.4f} x steps (fit over steps >= {int(n[2])});  residuals {np.round(t-(a+b*n),1).tolist()}")
PY
cat $O

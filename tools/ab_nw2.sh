#!/bin/bash
# GPU box: the two-scale loss with 2-wavefront workgroups for row blocks of up to R points (GLHIP_FWD_NW2_ROWS=R; 0: never)
for n in 3e4 1e5 2e5; do
  for r in 0 64 128; do
    echo -n "N = $n R = $r: "; GEOMLOSS_HIP_DENSE_SWITCH=0 GLHIP_FWD_NW2_ROWS=$r python tools/run_ms.py $n 3 2>/dev/null | tail -1
  done
done

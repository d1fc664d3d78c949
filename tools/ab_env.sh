#!/bin/bash
# usage: tools/ab_env.sh VAR "valA" "valB" file.hip -> rebuild file.hip with VAR=val (tools/build.py reads it) and bench twice each
V="$1"; A="$2"; B="$3"; F="$4"
for r in 1 2; do
  for v in "$A" "$B"; do
    touch pnnp_amd/csrc/$F
    env "$V=$v" python tools/build.py > /dev/null 2>&1
    echo "[$V=$v] $(python bench.py --no-kernel-events 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')"
  done
done

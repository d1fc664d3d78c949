#!/bin/bash
# usage: tools/ab_build.sh "<hipcc flags A>" "<hipcc flags B>" file.hip [bench args]: rebuilds file.hip with each flag set ON THE GPU BOX and runs bench.py twice each, interleaved
A="$1"; B="$2"; F="$3"; shift 3
for r in 1 2; do
  for v in "$A" "$B"; do
    touch pnnp_amd/csrc/$F
    PNNP_HIPCC_EXTRA="$v" python tools/build.py > /dev/null 2>&1
    echo "[$v] $(python bench.py --no-kernel-events "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')"
  done
done

#!/bin/bash
# usage: tools/ab_files.sh target.hip fileA fileB  -> builds with each file copied over pnnp_amd/csrc/target.hip, bench twice each (interleaved); restores fileB at the end
T="$1"; A="$2"; B="$3"
for r in 1 2; do
  for v in "$A" "$B"; do
    cp "$v" pnnp_amd/csrc/$T
    python tools/build.py > /dev/null 2>&1
    echo "[$v] $(python bench.py --no-kernel-events 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')"
  done
done

#!/bin/bash
# usage (on the GPU box): tools/ab_layer.sh file.hip "<flags A>" "<flags B>" ... -- <layer_bench args>
# rebuilds file.hip with each flag set and runs tools/layer_bench.py, twice, interleaved (box-to-box and run-to-run variance is several %)
F="$1"; shift
VARS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARS+=("$1"); shift; done
shift
for r in 1 2; do
  for v in "${VARS[@]}"; do
    touch pnnp_amd/csrc/$F
    PNNP_HIPCC_EXTRA="$v" python tools/build.py > /dev/null 2>&1
    echo "== [$v]"
    python tools/layer_bench.py "$@" 2>/dev/null | grep -v "^layer"
  done
done
touch pnnp_amd/csrc/$F; python tools/build.py > /dev/null 2>&1

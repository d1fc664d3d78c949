#!/bin/bash
# usage (on the GPU box): tools/ab_variants.sh "<tag1> <tag2> ..." <rounds> [layer_bench args]
# Interleaved A/B of pre-built library variants (tools/build_variant.sh <tag> ...: tools/scratch/variants/libpnnp_<tag>.so; the tag `base` is
# pnnp_amd/libpnnp_hip.so): per round and variant the selected layer_bench rows and one bench.py line.  Prints the sha256 of every binary first, so
# that two arms that are the SAME binary cannot pass for an A/B (ab_producer_two_register_sets.txt).
TAGS="$1"; ROUNDS="${2:-3}"; shift 2
lib_of() { if [ "$1" = base ]; then echo pnnp_amd/libpnnp_hip.so; else echo tools/scratch/variants/libpnnp_$1.so; fi; }
for t in $TAGS; do echo "binary [$t] $(sha256sum $(lib_of $t) | cut -c1-16)  $(lib_of $t)"; done
for r in $(seq 1 $ROUNDS); do
  for t in $TAGS; do
    echo "== round $r [$t]"
    if [ $# -gt 0 ]; then PNNP_LIB=$(lib_of $t) python tools/layer_bench.py "$@" 2>/dev/null | grep -v "^layer"; fi
    if [ -z "$AB_NO_STEP" ]; then
      PNNP_LIB=$(lib_of $t) python bench.py --no-kernel-events --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("  step", round(d["value"],2), "crops/s", round(d["ms_per_step"],3), "ms")'
    fi
  done
done

#!/bin/bash
# conv_x3s: start-up phase skew between workgroups (epilogue store bursts no longer coincide)
O=gpurun_out/r4e42; mkdir -p $O
(export PNNP_LIB=tools/scratch/variants/libpnnp_skew3000st.so
for a in "256 64 64 fwd" "256 64 64 dgrad" "512 32 32 fwd"; do python tools/x3s_stamps.py $a 2>&1 | grep -v "^/opt" >> $O/stamps.txt; done)
for v in new skew1500 skew3000 skew6000; do
  if [ $v = new ]; then unset PNNP_LIB; else export PNNP_LIB=tools/scratch/variants/libpnnp_$v.so; fi
  echo "== $v" >> $O/layers.txt
  python tools/layer_bench.py --x3 --only fwd,dgrad 2>&1 | grep -v "^/opt" >> $O/layers.txt
  echo "[$v config3] $(timeout 300 python bench.py --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), round(d["ms_per_step"],3))')" >> $O/bench_ab.txt
done
unset PNNP_LIB
cat $O/bench_ab.txt $O/stamps.txt; grep -E "==|total" $O/layers.txt

#!/bin/bash
# Round-6 GPU suite (run on the GPU box from the repo root):  bash tools/r6_suite.sh <tag> <commit>
# ORDER MATTERS: the PMC traffic passes come FIRST and their result goes to profiles/traffic.json on the box, so that the bench line written
# afterwards carries `roofline.traffic` measured on exactly these kernel sources (csrc_sha).
TAG=${1:-a}; COMMIT=${2:-unknown}
OUT=/root/repo/gpurun_out/r6s$TAG; mkdir -p $OUT
cd /root/repo
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
# 1. HBM traffic of the conv kernels (two --pmc passes) -> profiles/traffic.json
timeout 900 bash tools/pmc_traffic.sh $COMMIT > $OUT/pmc.log 2>&1
cp gpurun_out/traffic.json gpurun_out/pmc_fetch_size.csv gpurun_out/pmc_write_size.csv $OUT/ 2>/dev/null
cp gpurun_out/traffic.json profiles/traffic.json
# 2. tests
if [ -z "$SKIP_TESTS" ]; then
PNNP_SOAK_OUT=$OUT/soak.txt timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|ERROR" > $OUT/pytest.log
timeout 400 python -m pytest tests/test_gpu_h2.py tests/test_gpu_x3.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py -q -m gpu -s -k "accurate or golden or cancellation or dynamic_range or below or signed or tiny or wide_range or trajectory" 2>&1 | grep -i "relative L2\|worst HIP\|vs float64\|cancellation\|scale x\|below the\|signed mean\|convT wgrad\|x at 1e-36\|one channel\|over 40 steps" > $OUT/accuracy.log
fi
# 3. the bench line (default flags: h2 family, complete CPU-baseline protocol) and the rocprofv3 kernel statistics of the same command
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp_*
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_b -o b --output-format csv -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_bench.err
cp $(find /tmp/rp_b -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
cd /root/repo
timeout 600 python bench.py --steps 20 --warmup 5 --family x3 --no-cpu-baseline > $OUT/bench_family_x3.json 2> /dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --family wino --no-cpu-baseline > $OUT/bench_family_wino.json 2> /dev/null
# 4. config 5 (ResUnet + NoiseFlow proxy, B=12)
timeout 600 python bench.py --arch resunet --noise noiseflow --batch 12 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config5.json 2> $OUT/bench_config5.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_c5 -o c --output-format csv -- python3 /root/repo/bench.py --arch resunet --noise noiseflow --batch 12 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_under_rocprof.json 2> /dev/null
cp $(find /tmp/rp_c5 -name "*kernel_stats.csv" | head -1) $OUT/bench_config5_kernel_stats.csv
cd /root/repo
# 5. matrix-pipe utilisation per layer (PMC), with the co-execution and wait counters; LDS bank conflicts
timeout 900 bash tools/pmc_layers.sh util 'SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY' --h2 --only fwd,dgrad,wgrad --reps 2 > /dev/null 2>&1
cp gpurun_out/pmc_layers_util.csv $OUT/ 2>/dev/null
python tools/pmc_busy.py gpurun_out/pmc_layers_util.csv > $OUT/pmc_busy.txt 2>&1
timeout 900 bash tools/pmc_layers.sh lds 'SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU' --h2 --only fwd,dgrad,wgrad --reps 2 > /dev/null 2>&1
cp gpurun_out/pmc_layers_lds.csv $OUT/ 2>/dev/null
python tools/pmc_busy.py gpurun_out/pmc_layers_lds.csv > $OUT/pmc_lds.txt 2>&1
# 6. the smaller tables
timeout 900 bash tools/aux_prof.sh > $OUT/aux.log 2>&1
cp gpurun_out/aux_kernels.json gpurun_out/aux_kernel_stats.csv $OUT/ 2>/dev/null
timeout 300 python tools/eval_bench.py > $OUT/eval_bench.txt 2>&1
timeout 300 python tools/layer_bench.py --h2 > $OUT/layer_bench.txt 2>&1
timeout 300 python tools/layer_bench.py --x3 > $OUT/layer_bench_x3.txt 2>&1
timeout 300 python tools/pointwise_bench.py --h2 > $OUT/pointwise_bench.txt 2>&1
timeout 300 python tools/pointwise_bench.py > $OUT/pointwise_bench_x3.txt 2>&1
timeout 300 python tools/convt_wgrad_bench.py > $OUT/convt_wgrad_bench.txt 2>&1
timeout 200 python tools/squat_test.py 32 > $OUT/squat_test.txt 2>&1
./tools/ubench/h2_probe > $OUT/h2_probe.txt 2>&1
# 7. round 6: one 512 x 512 crop (split-K) per kernel, the corner tests of the fp16x2 family with their numbers
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp_one
printf 'import sys\nsys.path.insert(0, "/root/repo")\nimport torch\nfrom pnnp_amd.archs import UNetSeeInDark, initialize_weights\ntorch.manual_seed(0)\nnet = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda().eval()\nx = torch.rand(1, 4, 512, 512, device="cuda")\nwith torch.no_grad():\n    for _ in range(50): net(x)\ntorch.cuda.synchronize()\n' > /tmp/one_crop.py
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_one -o e --output-format csv -- python3 /tmp/one_crop.py > /dev/null 2>&1
cp $(find /tmp/rp_one -name "*kernel_stats.csv" | head -1) $OUT/one_crop_kernel_stats.csv
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_h2.py tests/test_gpu_dp2.py tests/test_gpu_unet.py -q -s -k "outlier or wide_range or shard or splitk or fused_head" 2>&1 | grep -i "rel L2\|split-K\|fused head\|shard\|passed\|failed" > $OUT/h2_corner_tests.txt
ls -la $OUT

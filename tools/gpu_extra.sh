#!/bin/bash
# Runs on the GPU box (via gpurun): the secondary bench lines (BASELINE configs[3], configs[4] at one GPU, float32 parity mode), a
# kernel trace of the configs[3] step, and the 2-rank control-flow rehearsal over gloo.   outputs under gpurun_out/<tag>/
set -o pipefail
tag=${1:-extra}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --extra --no-cpu-baseline > $out/extra.json 2> $out/extra.err || { tail -20 $out/extra.err; exit 1; }
python tools/show_bench.py $out/extra.json extra
P="--no-cpu-baseline --serial-streams --graph off --no-collapsed-line"
rocprofv3 --kernel-trace --stats -d $out/prof_c3 -o run -- python3 bench.py --steps 10 --warmup 3 --codebook 8192 --emb-dim 128 $P > $out/prof_c3.log 2>&1 || { tail -20 $out/prof_c3.log; exit 1; }
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline > $out/gloo2.json 2> $out/gloo2.err || { tail -20 $out/gloo2.err; exit 1; }
python tools/show_bench.py $out/gloo2.json value ms_per_step host_ms_per_step rank_ms_per_step
find $out -name '*kernel_trace.csv' -size +20M -delete
echo extra-ok

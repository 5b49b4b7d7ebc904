#!/bin/bash
# rocprofv3 kernel traces of the two auxiliary benches (pair mining, tile ingest), summarised on the box into small tables.
# usage (via gpurun): bash tools/aux_profiles.sh <tag>      outputs gpurun_out/<tag>_aux/*.md
set -o pipefail
tag=${1:-run}; out=gpurun_out/${tag}_aux; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_pairs -o run -- python3 tools/pairs_bench.py > $out/pairs.log 2>&1 || { tail -5 $out/pairs.log; exit 1; }
python3 tools/kernel_stats_small.py /tmp/prof_pairs/run_results.db $out/pairs_kernel_stats.md \
  "rocprofv3 --kernel-trace --stats -- python3 tools/pairs_bench.py (MI355X; 2000 / 4800 / 8000 anchors, D = 64, k = 8 / 16 / 8: avg over the three sizes, min = 2000, max = 8000 anchors)" knn_ > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_loader -o run -- python3 tools/loader_bench.py --epochs 4 > $out/loader.log 2>&1 || { tail -5 $out/loader.log; exit 1; }
python3 tools/kernel_stats_small.py /tmp/prof_loader/run_results.db $out/loader_kernel_stats.md \
  "rocprofv3 --kernel-trace --stats -- python3 tools/loader_bench.py --epochs 4 (MI355X; 256 tiles of 5x32x32x64 per launch, float16 -> bf16)" normalize_tiles > /dev/null
cat $out/*.md

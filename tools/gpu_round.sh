#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the bench line, a rocprofv3 kernel-trace summary and the two PMC passes.
# usage: tools/gpu_round.sh <tag> [quick|notests]     outputs under gpurun_out/<tag>/
set -o pipefail
tag=${1:-run}; quick=${2:-}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "$quick" != notests ]; then
  python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -2 $out/tests.log
fi
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
cat $out/bench.json
[ "$quick" = quick ] && exit 0
# profiled runs: eager launches (--graph off: one dispatch per kernel on the training streams) with the phase branch on the main stream
P="--no-cpu-baseline --serial-streams --graph off"
rocprofv3 --kernel-trace --stats -d $out/prof -o run -- python3 bench.py --steps 20 --warmup 5 $P > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o run -- python3 bench.py --steps 2 --warmup 1 --no-kernel-timing $P > $out/pmc_fetch.log 2>&1 || { tail -20 $out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o run -- python3 bench.py --steps 2 --warmup 1 --no-kernel-timing $P > $out/pmc_write.log 2>&1 || { tail -20 $out/pmc_write.log; exit 1; }
# matrix-core utilisation per kernel: MFMA busy cycles (summed over the 1024 SIMDs) against the kernel's own cycles (GRBM_GUI_ACTIVE / 8 XCDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_sq -o run -- python3 bench.py --steps 2 --warmup 1 --no-kernel-timing $P > $out/pmc_sq.log 2>&1 || { tail -20 $out/pmc_sq.log; exit 1; }
find $out -name '*kernel_trace.csv' -size +20M -delete
# the summaries (what gets committed under profiles/) are made here, next to the databases; gpurun brings back at most 64 MiB, so the raw
# databases stay on the box and only rocprofv3's own --stats tables travel with the summaries
python tools/summarize_profiles.py $out $tag > $out/summary.log 2>&1 || { tail -20 $out/summary.log; exit 1; }
mkdir -p $out/summary && cp profiles/${tag}_* $out/summary/
find $out/prof -name '*stats*.csv' -exec cp {} $out/summary/ \; 2>/dev/null
rm -rf $out/prof $out/pmc_fetch $out/pmc_write $out/pmc_sq
echo round-ok

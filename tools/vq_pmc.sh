#!/bin/bash
# VQ assignment micro-benchmark under rocprofv3: kernel time and HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of one library build.
# usage (via gpurun): bash tools/vq_pmc.sh <outdir> [lib tag]
set -o pipefail
out=$1; tag=${2:-}
mkdir -p $out; export TMPDIR=/tmp
[ -n "$tag" ] && export FRL_HIP_LIB_TAG=$tag
export VQ_BENCH_TILES=-1
python tools/vq_bench.py > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d $out/f -o run -- python3 tools/vq_bench.py > $out/f.log 2>&1 || { tail -5 $out/f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $out/w -o run -- python3 tools/vq_bench.py > $out/w.log 2>&1 || { tail -5 $out/w.log; exit 1; }
python3 - <<PY
import sqlite3, json
res = {}
for nm, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    c = sqlite3.connect("$out/%s/run_results.db" % nm)
    for k, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (cn,)):
        if "vq_assign" in k: res.setdefault(k[:40], {})[cn + "_KB"] = round(avg, 1)
for k, v in res.items():
    v["hbm_MB"] = round((2 * v.get("FETCH_SIZE_KB", 0) + v.get("WRITE_SIZE_KB", 0)) * 1024 / 1e6, 1)
print(json.dumps(res))
open("$out/pmc.json", "w").write(json.dumps(res, indent=1))
PY
grep stream_tiles $out/bench.log | cut -c1-140
rm -rf $out/f $out/w

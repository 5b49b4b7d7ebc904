"""Prints the headline and the per-kernel table of a bench.py JSON line: python tools/show_bench.py gpurun_out/x.json [n | key ...]"""
import json
import sys

d = json.load(open(sys.argv[1]))
rest = sys.argv[2:]
if rest and not rest[0].isdigit():                    # named keys: print them as JSON
    for k in rest:
        print(k, json.dumps(d.get(k), indent=1))
    sys.exit(0)
n = int(rest[0]) if rest else 100
gk = d.get("gpu_kernels", {})
print(d["value"], "tiles/s", d["ms_per_step"], "ms/step; kernels", d.get("gpu_kernel_ms_per_step"), "ms; launches", sum(v["launches_per_step"] for v in gk.values()),
      "perplexity", d["config"].get("perplexity"))
for k, v in list(gk.items())[:n]:
    print(f'{v["ms_per_step"]:.4f} {v["launches_per_step"]:5.1f} {v["avg_us"]:8.1f}  {k[:100]}')

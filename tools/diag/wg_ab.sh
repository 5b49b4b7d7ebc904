timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "conv1x1 or bwd_weight or wgrad or pointwise or film or phase" > gpurun_out/t_wg.log 2>&1; tail -3 gpurun_out/t_wg.log; for m in 1024 512 256; do timeout -k 10 100 python tools/wgrad_bench.py --max-wgs $m > gpurun_out/wgrad_$m.json 2>/dev/null; done; python - <<EOF
import json
for m in (1024,512,256):
    d=json.load(open("gpurun_out/wgrad_%d.json"%m))
    print(m, {k[:22]:(v["call_us"], v["kernels_us"].get("pw_wgrad_kernel")) for k,v in d.items() if "pw_wgrad_kernel" in v["kernels_us"]})
EOF

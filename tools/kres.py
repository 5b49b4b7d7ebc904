#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

Usage: python tools/kres.py vq-vae_amd/csrc/tcn_hot.hip [substring-filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "vq-vae_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for c in rows:
        if flt and flt not in c["name"]:
            continue
        name = subprocess.run(["/usr/bin/c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:70]
        print(f"{name:70s} {c.get('VGPRs', '?'):>5s} {c.get('AGPRs', '?'):>5s} {c.get('SGPRs', '?'):>5s} "
              f"{c.get('ScratchSize [bytes/lane]', '?'):>8s} {c.get('LDS Size [bytes/block]', '?'):>7s} {c.get('Occupancy [waves/SIMD]', '?'):>4s}")


if __name__ == "__main__":
    main()

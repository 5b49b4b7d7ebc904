#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/gpu_round.sh (rocpd sqlite databases under gpurun_out/<tag>/) into the committed
summaries under profiles/:  <name>_kernel_stats.{md,csv}  and  <name>_pmc.json  (+ the bench JSON line).

usage: tools/summarize_profiles.py gpurun_out/<tag> <name>          e.g.  gpurun_out/r01b r01b

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes,
both are in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it is doubled.
"""
import json
import os
import shutil
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                     "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    return [dict(name=r[0], calls=r[1], total_ms=r[2] / 1e6, avg_us=r[3] / 1e3, min_us=r[4] / 1e3, max_us=r[5] / 1e3, pct=100.0 * r[2] / tot,
                 vgpr=r[6], agpr=r[7], sgpr=r[8], lds=r[9], scratch=r[10], grid=r[11], wg=r[12]) for r in rows], tot / 1e6


def pmc(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name",
                                  (counter,)):
        out[name] = (n, avg)
    return out


def main():
    src, name = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    stats, tot = kernel_stats(os.path.join(src, "prof", "run_results.db"))
    steps = max([st["calls"] for st in stats if "frl_adamw_kernel" in st["name"]] or [30])      # train steps in the profiled process
    cmd = "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --serial-streams --graph off"
    with open(f"profiles/{name}_kernel_stats.csv", "w") as f:
        f.write("name,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds,scratch,grid,wg\n")
        for s in stats:
            f.write('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.3f,%d,%d,%d,%d,%d,%d,%d\n' % (s["name"], s["calls"], s["total_ms"], s["avg_us"], s["min_us"],
                                                                                s["max_us"], s["pct"], s["vgpr"], s["agpr"], s["sgpr"], s["lds"],
                                                                                s["scratch"], s["grid"], s["wg"]))
    with open(f"profiles/{name}_kernel_stats.md", "w") as f:
        f.write(f"# {cmd}  (MI355X, {name})\n\n{steps} train steps in the process (warm-up, timed, event-instrumented and the secondary collapsed-codebook line), phase branch on the main stream so that kernel durations are not stretched by the overlap of the two branches; GPU kernel time total "
                f"{tot:.1f} ms = {tot / steps:.3f} ms/step.\n\n| kernel | calls | total ms | avg us | % | vgpr | lds B | scratch B |\n|---|---|---|---|---|---|---|---|\n")
        for s in stats[:60]:
            f.write("| `%s` | %d | %.2f | %.1f | %.2f | %d | %d | %d |\n" % (s["name"][:90], s["calls"], s["total_ms"], s["avg_us"], s["pct"],
                                                                           s["vgpr"], s["lds"], s["scratch"]))
    fe = pmc(os.path.join(src, "pmc_fetch", "run_results.db"), "FETCH_SIZE")
    wr = pmc(os.path.join(src, "pmc_write", "run_results.db"), "WRITE_SIZE")
    ks = {}
    for k in sorted(set(fe) | set(wr)):
        f_kb = fe.get(k, (0, 0.0))[1]
        w_kb = wr.get(k, (0, 0.0))[1]
        ks[k] = {"launches": fe.get(k, wr.get(k))[0], "FETCH_SIZE_KB_avg": round(f_kb, 2), "WRITE_SIZE_KB_avg": round(w_kb, 2),
                 "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024)}
    json.dump({"note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                       "--no-kernel-timing --serial-streams`; values in KiB averaged over the launches of each kernel; hbm_bytes_per_launch = (2*FETCH_SIZE + "
                       "WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md)",
               "kernels": ks}, open(f"profiles/{name}_pmc.json", "w"), indent=1)
    sqdb = os.path.join(src, "pmc_sq", "run_results.db")
    if os.path.exists(sqdb):
        names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
        per = {n: pmc(sqdb, n) for n in names}
        out = {}
        for k in sorted(set().union(*[set(v) for v in per.values()])):
            row = {n: per[n][k][1] for n in names if k in per[n]}
            if row.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
                cycles = row["GRBM_GUI_ACTIVE"] / 8.0                       # the counter is the sum over the 8 XCDs
                row["kernel_cycles"] = round(cycles, 1)
                row["mfma_util"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 4)      # 256 CUs x 4 SIMDs
                if row.get("SQ_WAVE_CYCLES", 0) > 0:
                    row["wave_time_waiting"] = round(row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"], 3)
            out[k] = {kk: (round(vv, 4 if kk in ("mfma_util", "wave_time_waiting") else 1) if isinstance(vv, float) else vv) for kk, vv in row.items()}
        conv = {}
        for fam, subs in (("tcn_hot_bwd", ("tcn_hot_bwd",)), ("tcn_hot_fwd", ("tcn_hot_fwd",)), ("tcn_chain_fwd", ("tcn_chain_fwd",)),
                          ("conv3x3", ("conv3x3_kernel",)), ("conv3x3_wgrad", ("conv3x3_wgrad",)), ("pw_conv", ("pw_conv_kernel",)),
                          ("pw_wgrad", ("pw_wgrad",)), ("dec_mse", ("dec_mse_",)), ("enc2", ("enc2_",)), ("smooth_heads", ("smooth_heads",)),
                          ("film_fused", ("film_fused",)), ("vq_assign", ("vq_assign",))):
            hits = [v for k, v in out.items() if any(s in k for s in subs) and "mfma_util" in v]
            if hits:
                busy = sum(h["SQ_VALU_MFMA_BUSY_CYCLES"] for h in hits)
                cyc = sum(h["kernel_cycles"] for h in hits)
                conv[fam] = round(busy / (cyc * 1024.0), 4)
        json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                           "-- python3 bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline --serial-streams --graph off; averages per "
                           "launch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of SIMD cycles with the "
                           "matrix pipe busy (16 cycles per v_mfma_f32_16x16x32_bf16)",
                   "kernels": conv, "per_kernel": out}, open(f"profiles/{name}_mfma_util.json", "w"), indent=1)
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), f"profiles/{name}_bench.json")
    print(f"wrote profiles/{name}_kernel_stats.md/.csv, profiles/{name}_pmc.json; total {tot / steps:.3f} ms/step over {steps} steps")
    for s in stats[:12]:
        print("%-70s %5d %8.1f us %6.2f%%" % (s["name"][:70], s["calls"], s["avg_us"], s["pct"]))


if __name__ == "__main__":
    main()

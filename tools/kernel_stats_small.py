#!/usr/bin/env python3
"""rocpd sqlite (rocprofv3 --kernel-trace) -> small markdown table of the kernels whose names match the given substrings.
usage: tools/kernel_stats_small.py <run_results.db> <out.md> <title> <substr> [<substr> ...]"""
import sqlite3
import sys

db, out, title, subs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), max(scratch_size), "
                 "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
with open(out, "w") as f:
    f.write(f"# {title}\n\n| kernel | calls | avg us | min us | max us | vgpr | lds B | scratch B | grid | wg |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        if any(s in r[0] for s in subs):
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %d | %d | %d | %d | %d |\n" % (r[0][:110], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7], r[8], r[9]))
print(open(out).read())

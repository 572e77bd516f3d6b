#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) as a per-kernel table
(calls, total/avg/min/max us, % of GPU kernel time, grid, VGPR/LDS) -- what --stats prints."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                    max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(lds_size)
                    from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"# rocprofv3 --kernel-trace --stats summary of {db.split('/')[-1]}; durations in us")
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}")
for n, cnt, s, a, mn, mx, g, wg, v, av, lds in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n[:90]:90s} {cnt:6d} {s/1e3:11.1f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f} {g:8d} {wg:5d} {v:5d} {av:5d} {lds:7d}")

#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`
on ROCm 7.2) as a per-kernel table: calls, total / mean / min / max duration (us), grid, LDS,
VGPR.  Usage: python tools/rocpd_summary.py <results.db> [--skip-first N] > profiles/xyz.txt"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, "
                      "accum_vgpr_count, sgpr_count, start from kernels order by start").fetchall()
    by = {}
    for name, dur, gx, wx, lds, vg, ag, sg, _ in rows:
        by.setdefault(name, []).append((dur, gx, wx, lds, vg, ag, sg))
    total = sum(sum(d[0] for d in v[skip:]) for v in by.values()) or 1
    print(f"# {path}: {len(rows)} kernel dispatches; first {skip} dispatches of each kernel skipped")
    print(f"{'calls':>6} {'total_us':>10} {'mean_us':>9} {'min_us':>8} {'max_us':>8} {'%':>6} "
          f"{'grid':>9} {'wg':>5} {'lds':>7} {'vgpr':>5} {'sgpr':>5}  kernel")
    for name, v in sorted(by.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
        w = v[skip:] or v
        durs = [d[0] / 1e3 for d in w]
        _, gx, wx, lds, vg, ag, sg = w[-1]
        print(f"{len(durs):6d} {sum(durs):10.1f} {sum(durs)/len(durs):9.2f} {min(durs):8.2f} "
              f"{max(durs):8.2f} {100*sum(d[0] for d in w)/total:6.1f} {gx:9d} {wx:5d} {lds:7d} "
              f"{vg + ag:5d} {sg:5d}  {name[:110]}")


if __name__ == "__main__":
    main()

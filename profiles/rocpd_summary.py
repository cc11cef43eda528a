#!/usr/bin/env python3
"""Per-kernel summary (count, avg / min / max duration, share) of a rocprofv3 --kernel-trace run
stored as a rocpd SQLite database (rocprofv3's default output on ROCm 7.2).

Register columns: rocpd's arch_vgpr_count is half the wave64 allocation on gfx950 (it reads 112 for a kernel the
compiler reports with 222 VGPRs, allocated as 224; 120 for 238 -> 240; 76 for 146 -> 152), so the `vgpr` column
printed here is 2 x arch_vgpr_count = registers allocated per lane; `agpr` is rocpd's accum_vgpr_count as it is.

    python profiles/rocpd_summary.py gpurun_out/prof_x/name_results.db > profiles/summary.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(
        "select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
        "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 6 desc" % (disp, sym)))
    total = sum(r[5] for r in rows) or 1
    print("%-72s %7s %11s %11s %11s %11s %6s %5s %5s %5s %7s %9s %5s" % (
        "kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct", "vgpr", "agpr", "sgpr", "lds_B", "grid_x", "wg_x"))
    for r in rows:
        print("%-72s %7d %11.2f %11.2f %11.2f %11.3f %6.2f %5d %5d %5d %7d %9d %5d" % (
            r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6, 100.0 * r[5] / total, 2 * r[6], r[7], r[8], r[9], r[10], r[11]))


if __name__ == "__main__":
    main(sys.argv[1])

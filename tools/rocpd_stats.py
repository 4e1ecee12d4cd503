#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats table
(calls, total / average / min / max duration), the same content as `--stats -f csv`."""

import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    q = """
    select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
           max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),
           max(d.group_segment_size)
    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
    group by s.kernel_name order by 3 desc
    """
    rows = db.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,GridX,WorkgroupX,ArchVGPR,AccumVGPR,SGPR,LDSBytes"]
    for r in rows:
        lines.append(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / total:.2f},{r[4]},{r[5]},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}')
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

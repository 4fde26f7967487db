#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, share) from a rocprofv3 rocpd .db — the same
table `rocprofv3 --stats` prints as kernel_stats.csv.  Usage: rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(
    f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
    f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
    f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ['Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS_bytes,GridSizeX,WorkgroupSizeX']
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total,
                                                               r[6], r[7], r[8], r[9], r[10]))
text = '\n'.join(lines) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text)
print(text)

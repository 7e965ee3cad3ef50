#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 (rocpd sqlite) kernel trace: calls, total / average / min / max duration.
usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = {r[0].split("_0000")[0]: r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    kd, ks = tabs["rocpd_kernel_dispatch"], tabs["rocpd_info_kernel_symbol"]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    q = f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = cur.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[2] / r[1], r[3], r[4], 100.0 * r[2] / tot) for r in rows], cols


if __name__ == "__main__":
    rows, _ = stats(sys.argv[1])
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for r in rows:
        nm = r[0].replace("(anonymous namespace)::", "").split("(")[0]
        lines.append(f"\"{nm}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{r[6]:.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)

#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls, total,
average, share.  Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_us,avg_us,min_us,max_us,percent"]
    for name, n, tot, mn, mx in rows:
        short = name.split("(")[0][:90].replace(",", ";")
        lines.append(f"{short},{n},{tot / 1e3:.1f},{tot / n / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * tot / total:.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()

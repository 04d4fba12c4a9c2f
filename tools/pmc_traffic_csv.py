#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (rocpd .db):
FETCH_SIZE (KB, doubled per the gfx950 note of MI355X_MICROARCH.md "HBM") and
WRITE_SIZE (KB).  Usage: python tools/pmc_traffic_csv.py fetch.db write.db [out.csv]"""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]  # noqa: E731
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    ev = "event_id" if "event_id" in kcols else "id"
    q = (f"select s.{name_col}, d.id, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.{ev} join {ks} s on d.kernel_id = s.id where p.name = ? group by d.id")
    acc = defaultdict(list)
    for name, _, val in cur.execute(q, (counter,)):
        acc[name.split("(")[0][:90].replace(",", ";")].append(val)
    return acc


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in f:
        fk = sum(f[k]) / len(f[k])
        wk = sum(w[k]) / len(w[k]) if k in w and w[k] else 0.0
        fb, wb = fk * 1024 * 2, wk * 1024
        rows.append((len(f[k]) * (fb + wb), f"{k},{len(f[k])},{fk:.1f},{int(fb)},{wk:.1f},{int(wb)},{int(fb + wb)}"))
    rows.sort(reverse=True)
    out = "\n".join(["kernel,launches,FETCH_SIZE_KB_per_launch(raw),fetch_bytes_per_launch(x2 gfx950 correction),"
                     "WRITE_SIZE_KB_per_launch,write_bytes_per_launch,hbm_bytes_per_launch"] + [r[1] for r in rows])
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out + "\n")


if __name__ == "__main__":
    main()

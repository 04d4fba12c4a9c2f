#!/usr/bin/env python3
"""Per-kernel PMC summary from a rocprofv3 rocpd .db: for every kernel name the
per-dispatch mean of each collected counter (summed over instances)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]  # noqa: E731
    kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    ev = "event_id" if "event_id" in kcols else "id"
    q = (f"select s.{name_col}, d.id, p.name, sum(e.value), d.end - d.start, d.grid_size_x, d.grid_size_y, d.grid_size_z "
         f"from {pe} e join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.{ev} "
         f"join {ks} s on d.kernel_id = s.id group by d.id, p.name")
    acc = defaultdict(lambda: defaultdict(list))
    for name, did, pname, val, dur, gx, gy, gz in cur.execute(q):
        key = name.split("(")[0][:70] + f" grid={gx}x{gy}x{gz}"
        acc[key][pname].append(val)
        acc[key]["_dur_us"].append(dur / 1e3)
    for key, d in acc.items():
        n = len(d["_dur_us"]) // max(1, (len(d) - 1))
        parts = [f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(d.items())]
        print(key, "n=%d" % len(next(iter(d.values()))), " ".join(parts))


if __name__ == "__main__":
    main()

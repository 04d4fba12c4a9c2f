#!/usr/bin/env python3
"""Per-kernel matrix-core / LDS counters from a rocprofv3 --pmc pass (rocpd .db) of
    SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
Per kernel name: launches, mean duration, the per-dispatch mean of every counter (summed over instances) and
mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) - the share of all SIMD cycles of the chip
during which the matrix pipe was busy (MI355X_MICROARCH.md: the counter counts cycles, 32 per v_mfma_f32_32x32x2 /
16x16x4 f32 instruction; SQ_WAIT_* count quad-cycles).
Usage: python tools/pmc_sq_csv.py sq.db [out.csv]"""
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
    q = (f"select s.{name_col}, d.id, p.name, sum(e.value), d.end - d.start from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.{ev} join {ks} s on d.kernel_id = s.id group by d.id, p.name")
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for name, did, pname, val, d in cur.execute(q):
        key = name.split("(")[0][:90].replace(",", ";")
        acc[key][pname].append(val)
        dur[key][did] = d / 1e3
    counters = sorted({c for d in acc.values() for c in d})
    rows = []
    for key, d in acc.items():
        n = len(dur[key])
        du = sum(dur[key].values()) / n
        means = {c: (sum(d[c]) / len(d[c]) if d.get(c) else 0.0) for c in counters}
        mf = means.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(du * 2400.0 * 1024.0, 1e-9) * 100.0
        rows.append((n * du, f"{key},{n},{du:.2f}," + ",".join(f"{means[c]:.4g}" for c in counters) + f",{mf:.2f}"))
    rows.sort(reverse=True)
    out = "\n".join(["kernel,launches,avg_us_under_pmc," + ",".join(counters) + ",mfma_busy_pct_of_all_simd_cycles"] + [r[1] for r in rows])
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()

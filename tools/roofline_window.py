#!/usr/bin/env python3
"""One table for the SAME launches: the roofline leg of bench.py (bracketed by sc_marker_kernel<0> / <1>) as seen by
  * bench.py itself        (--roofline-csv: launches, HIP-event time, ALGORITHMIC bytes / flops per launch),
  * rocprofv3 --kernel-trace        (launches and average duration inside the window),
  * rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes (HBM bytes per launch inside the window:
    FETCH_SIZE KB x 2 (the gfx950 correction of MI355X_MICROARCH.md "HBM") + WRITE_SIZE KB).
Every pass is its own run of the same command; a pass is cut to the dispatches that START between its two marker kernels
(all queues).  Usage: python tools/roofline_window.py bench_roofline.csv trace.db fetch.db write.db out.csv [note]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict

KINDS = [("gemm_naive_kernel", r"gemm_naive_kernel"), ("gemm_skinny_kernel", r"gemm_skinny_kernel"),
         ("gemm_mfma_kernel<128,128>", r"gemm_mfma_kernelILi128ELi128"), ("gemm_mfma_kernel<64,64>", r"gemm_mfma_kernelILi64ELi64"),
         ("proj_ln_proj_kernel<256,*>", r"(proj_ln_proj_kernelILi256|reduce_ln_proj_kernelILi256)"),
         ("ffn_fused_kernel<256,*>", r"ffn_fused_kernelILi256ELi\dELb0"), ("ffn_fused_kernel<256,*,PRO>", r"ffn_fused_kernelILi256ELi\dELb1"),
         ("dec_attn_flash_kernel<self>", r"dec_attn_flash_kernelILi32ELi10ELb1"), ("dec_attn_flash_kernel<cross>", r"dec_attn_flash_kernelILi32ELi10ELb0"),
         ("rowtile_proj_kernel<256,*>", r"rowtile_proj_kernelILi256"),
         ("dec_layer_attn_kernel<self>", r"dec_layer_attn_kernelILi256ELi32ELi10ELb1"),
         ("dec_layer_attn_kernel<cross>", r"dec_layer_attn_kernelILi256ELi32ELi10ELb0")]


def kind_of(name):
    for k, pat in KINDS:
        if re.search(pat, name):
            return k
    return None


def window(db, counter=None):
    """{kind: [n, total duration us, total counter]} of the dispatches between the markers; also per raw kernel name"""
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]  # noqa: E731
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = list(cur.execute(f"select d.id, s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"))
    m0 = [r[2] for r in rows if "sc_marker_kernelILi0" in r[1] or "sc_marker_kernel<0>" in r[1]]
    m1 = [r[2] for r in rows if "sc_marker_kernelILi1" in r[1] or "sc_marker_kernel<1>" in r[1]]
    if not m0 or not m1:
        raise SystemExit(f"{db}: marker kernels not found (bench.py must run its roofline leg: --roofline-steps > 0)")
    lo, hi = m0[-1], min(x for x in m1 if x > m0[-1])
    val = {}
    if counter:
        kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
        ev = "event_id" if "event_id" in kcols else "id"
        pe, pi = t("rocpd_pmc_event"), t("rocpd_info_pmc")
        q = (f"select d.id, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.{ev} "
             f"where p.name = ? group by d.id")
        val = dict(cur.execute(q, (counter,)))
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for did, name, st, en in rows:
        if not (lo < st < hi):
            continue
        k = kind_of(name) or ("other: " + name.split("(")[0][:60])
        a = acc[k]
        a[0] += 1
        a[1] += (en - st) / 1e3
        a[2] += val.get(did, 0.0)
    return acc, (hi - lo) / 1e3


def main():
    bench_csv, trace_db, fetch_db, write_db, out = sys.argv[1:6]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    with open(bench_csv) as f:
        bench = {r["kind"]: r for r in csv.DictReader(x for x in f if not x.startswith("#"))}
    tr, span = window(trace_db)
    fe, _ = window(fetch_db, "FETCH_SIZE")
    wr, _ = window(write_db, "WRITE_SIZE")
    with open(out, "w", newline="") as f:
        f.write("# the roofline leg of `python bench.py` (launches between sc_marker_kernel<0> and <1>) in four runs of the same command: "
                "bench.py's own table, rocprofv3 --kernel-trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE (tools/prof_bench.sh)\n")
        f.write(f"# window of the trace pass: {span:.0f} us; hbm_bytes_per_launch_pmc = FETCH_SIZE KB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KB x 1024. {note}\n")
        w = csv.writer(f)
        w.writerow(["kind", "launches_bench", "avg_launch_us_events", "algorithmic_bytes_per_launch", "algorithmic_mflop_per_launch",
                    "launches_trace", "avg_us_trace", "launches_pmc", "fetch_kb_per_launch_raw", "write_kb_per_launch",
                    "hbm_bytes_per_launch_pmc", "traffic_over_algorithmic", "algorithmic_gbs_trace", "frac_of_8000_gbs", "algorithmic_tflops_trace", "frac_of_157_3_tflops"])
        kinds = [k for k, _ in KINDS] + sorted(k for k in tr if k.startswith("other: "))
        for k in kinds:
            if k not in tr and k not in bench:
                continue
            b = bench.get(k, {})
            n_t, us_t, _ = tr.get(k, [0, 0.0, 0.0])
            n_f, _, v_f = fe.get(k, [0, 0.0, 0.0])
            n_w, _, v_w = wr.get(k, [0, 0.0, 0.0])
            avg_t = us_t / n_t if n_t else 0.0
            fkb, wkb = (v_f / n_f if n_f else 0.0), (v_w / n_w if n_w else 0.0)
            hbm = fkb * 1024 * 2 + wkb * 1024
            ab = float(b.get("algorithmic_bytes_per_launch", 0) or 0)
            mf = float(b.get("algorithmic_mflop_per_launch", 0) or 0)
            gbs = ab / (avg_t * 1e-6) / 1e9 if avg_t and ab else 0.0
            tfl = mf * 1e6 / (avg_t * 1e-6) / 1e12 if avg_t and mf else 0.0
            w.writerow([k, b.get("launches", ""), b.get("avg_launch_us_events", ""), int(ab) if ab else "", mf or "",
                        n_t, round(avg_t, 2), n_f, round(fkb, 1), round(wkb, 1), int(hbm) if n_f else "",
                        round(hbm / ab, 3) if ab and n_f else "", round(gbs, 1) if gbs else "", round(gbs / 8000.0, 4) if gbs else "",
                        round(tfl, 2) if tfl else "", round(tfl / 157.3, 4) if tfl else ""])
    print(open(out).read())


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where a queue of a rocprofv3 rocpd (.db) kernel trace sits idle: every gap > min_us between consecutive kernels of the decode
queue inside a window, with the kernel before and after it and what the OTHER queues ran during the gap (kernel time, names).
Usage: python tools/rocpd_gaps.py <results.db> <start_ms_from_end> <length_ms> [min_us=20]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    back, length = float(sys.argv[2]), float(sys.argv[3])
    min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in kcols else "stream_id"
    rows = list(cur.execute(f"select s.{name_col}, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    t_end = max(r[2] for r in rows)
    t0 = t_end - back * 1e6
    t1 = t0 + length * 1e6
    dq = [r[3] for r in rows if "dec_layer_" in r[0]]
    dq = max(set(dq), key=dq.count)
    dec = [r for r in rows if r[3] == dq and t0 <= r[1] < t1]
    oth = [r for r in rows if r[3] != dq and r[2] > t0 and r[1] < t1]
    short = lambda n: n.replace("_Z21dec_layer_attn_kernelILi256ELi32ELi10E", "dec_attn<").replace("_Z16ffn_fused_kernelILi256E", "ffn<").replace("_Z23dec_layer_stream_kernelI", "dec_stream<")[:40]
    busy = sum(r[2] - r[1] for r in dec)
    print(f"decode queue {dq}: {len(dec)} kernels, busy {busy / 1e3:.0f} us of {(t1 - t0) / 1e3:.0f} us; other queues: {len(oth)} kernels, {sum(min(r[2], t1) - max(r[1], t0) for r in oth) / 1e3:.0f} us")
    for a, b in zip(dec, dec[1:]):
        gap = (b[1] - a[2]) / 1e3
        if gap < min_us:
            continue
        during = [o for o in oth if o[2] > a[2] and o[1] < b[1]]
        ot = sum(min(o[2], b[1]) - max(o[1], a[2]) for o in during) / 1e3
        names = {}
        for o in during:
            names[short(o[0])] = names.get(short(o[0]), 0) + 1
        top = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
        print(f"{(a[2] - t0) / 1e3:9.1f} us  gap {gap:8.1f} us  after {short(a[0]):28s} before {short(b[0]):28s} | other queues busy {ot:7.1f} us: {top}")


main()

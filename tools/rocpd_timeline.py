#!/usr/bin/env python3
"""Timeline of one decode step from a rocprofv3 rocpd (.db) kernel trace: every
kernel between two dec_embed launches with its duration and the idle gap before
it.  Usage: python tools/rocpd_timeline.py <results.db> [which_step] [max_rows]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    which = sys.argv[2] if len(sys.argv) > 2 else "-3"      # a step index, "hpw4": the last step that starts with a four-heads-per-workgroup kernel, "full": of a full bucket
    max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in kcols else ("stream_id" if "stream_id" in kcols else None)
    gcol = "d.grid_size_x * d.grid_size_y / (d.workgroup_size_x * d.workgroup_size_y)" if "grid_size_x" in kcols and "workgroup_size_x" in kcols else "0"
    rows = list(cur.execute(f"select s.{name_col} || ' wgs=' || cast({gcol} as text), d.start, d.end, {('d.' + qcol) if qcol else '0'} from {kd} d "
                            f"join {ks} s on d.kernel_id = s.id order by d.start"))
    # only the decode stream's queue (continuous batching: encoder groups run beside it on another stream)
    import re
    first = re.compile(r"dec_layer_attn_kernelILi\d+ELi\d+ELi\d+ELb1ELi\d+ELb1ELb[01](ELi\d+)?(ELb[01])?EEv|dec_layer_stream_kernelILb1E")
    dq = [r[3] for r in rows if "dec_embed" in r[0] or first.search(r[0])]
    if dq:
        q = max(set(dq), key=dq.count)
        n_other = sum(1 for r in rows if r[3] != q)
        rows = [r[:3] for r in rows if r[3] == q]
        print(f"(decode queue {q}: {len(rows)} kernels; {n_other} kernels on other queues not shown)")
    else:
        rows = [r[:3] for r in rows]
    # a decode step starts with dec_embed (six-launch layers) or with the FIRST variant of the head-parallel
    # self-attention layer kernel (last template argument true)
    # mangled: dec_layer_attn_kernel<D, DK, WM, SELF = true, UNR, FIRST = true, KVH[, HPW[, WH]]>
    starts = [i for i, r in enumerate(rows) if "dec_embed" in r[0] or first.search(r[0])]
    if which == "hpw4":
        cand = [k for k, i in enumerate(starts[:-1]) if "ELi4ELb" in rows[i][0]]
        which = cand[-2] - len(starts) if len(cand) > 1 else -3
    if which == "full":   # the last step of a (nearly) full bucket: its decoder FFN runs 48-row tiles (RTT = 3)
        cand = [k for k, i in enumerate(starts[:-1])
                if any("ffn_fused_kernelILi256ELi3ELb1" in r[0] or
                       ("dec_layer_stream_kernel" in r[0] and int(r[0].rsplit("wgs=", 1)[1]) >= 100) for r in rows[i:starts[k + 1]])]
        which = cand[-2] - len(starts) if len(cand) > 1 else -3
    which = int(which)
    a, b = starts[which], starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
    seg = rows[a:b]
    busy = sum(r[2] - r[1] for r in seg)
    span = seg[-1][2] - seg[0][1]
    print(f"decode step #{which}: {len(seg)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us "
          f"({100.0 * busy / span:.1f} %)")
    prev_end = seg[0][1]
    agg = {}
    for i, (name, s, e) in enumerate(seg):
        short = name.split("(")[0].replace("void ", "")[:60] + (" " + name[name.rindex(" wgs="):] if " wgs=" in name else "")
        if i < max_rows:
            print(f"{(s - seg[0][1]) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  {short}")
        g = agg.setdefault(short, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += (e - s) / 1e3
        g[2] += (s - prev_end) / 1e3
        prev_end = e
    print("\nper kernel over the step: calls, total dur us, total gap-before us")
    for k, (n, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:4d} {dur:9.1f} {gap:8.1f}  {k}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Who has the GPU when: the kernel trace (rocprofv3 rocpd .db) of a continuous-batching run as a sequence of PHASES - a decode
iteration (queue of the decoder layer kernels: from a FIRST-layer self-attention kernel to the kernel before the next one) or
an encoder group (the other busy queue: from logmel_kernel to the kernel before the next logmel_kernel).  Per phase: start,
span, summed kernel time, streams (decode: workgroups of the layer kernel / heads groups; encoder: logmel workgroups), and
how much of its span overlaps kernels of the other kind.
Usage: python tools/rocpd_phases.py <results.db> [last_ms=200] [detail]"""
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    detail = len(sys.argv) > 3
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in kcols else "stream_id"
    rows = list(cur.execute(f"select s.{name_col}, d.start, d.end, d.{qcol}, d.grid_size_x * d.grid_size_y * d.grid_size_z / "
                            f"(d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z) from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    t_end = max(r[2] for r in rows)
    rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
    t0 = rows[0][1]
    # a decode iteration starts with the FIRST variant of a layer kernel (head-parallel or, round 6, stream-resident)
    first = re.compile(r"dec_layer_attn_kernelILi\d+ELi\d+ELi\d+ELb1ELi\d+ELb1|dec_layer_stream_kernelILb1E")
    hpw = re.compile(r"dec_layer_attn_kernelILi\d+ELi\d+ELi\d+ELb[01]ELi\d+ELb[01]ELb[01]ELi(\d+)")
    dq = [r[3] for r in rows if first.search(r[0])]
    eq = [r[3] for r in rows if "logmel_kernel" in r[0]]
    dq = max(set(dq), key=dq.count)
    eq = max(set(eq), key=eq.count) if eq else -1
    phases = []   # [kind, start, end, busy, streams, n]
    for name, s, e, q, wgs in rows:
        if q == dq:
            if first.search(name) or not phases or phases[-1][0] != "dec" and not any(p[0] == "dec" for p in phases[-2:]):
                if first.search(name):
                    m = hpw.search(name)
                    per = 1 if "dec_layer_stream" in name else (2 if m and m.group(1) == "4" else 8)
                    phases.append(["dec", s, e, 0.0, wgs // per, 0])
            cand = [p for p in phases if p[0] == "dec"]
            if cand:
                p = cand[-1]
                p[2] = max(p[2], e); p[3] += e - s; p[5] += 1
        elif q == eq:
            if "logmel_kernel" in name:
                phases.append(["enc", s, e, 0.0, wgs, 0])
            cand = [p for p in phases if p[0] == "enc"]
            if cand:
                p = cand[-1]
                p[2] = max(p[2], e); p[3] += e - s; p[5] += 1
    phases.sort(key=lambda p: p[1])
    tot = {"dec": 0.0, "enc": 0.0}
    ovl = 0.0
    for i, p in enumerate(phases):
        o = 0.0
        for qh in phases:
            if qh[0] != p[0]:
                o += max(0.0, min(p[2], qh[2]) - max(p[1], qh[1]))
        p.append(o)
        tot[p[0]] += p[2] - p[1]
        if p[0] == "enc":
            ovl += o
    wall = phases[-1][2] - phases[0][1]
    print(f"last {last_ms:.0f} ms of the trace: {sum(1 for p in phases if p[0] == 'dec')} decode iterations, {sum(1 for p in phases if p[0] == 'enc')} encoder groups, wall {wall / 1e6:.2f} ms")
    print(f"decode iteration spans {tot['dec'] / 1e6:.2f} ms ({100 * tot['dec'] / wall:.1f} %), encoder group spans {tot['enc'] / 1e6:.2f} ms ({100 * tot['enc'] / wall:.1f} %), "
          f"of which beside a decode iteration {ovl / 1e6:.2f} ms ({100 * ovl / max(tot['enc'], 1):.1f} % of the encoder spans)")
    import collections
    hs = collections.Counter(min(p[4] // 16 * 16, 128) for p in phases if p[0] == "dec")
    print("decode bucket sizes (streams, bins of 16):", dict(sorted(hs.items())))
    hs = collections.Counter(p[4] // 8 * 8 for p in phases if p[0] == "enc")
    print("encoder group sizes (logmel workgroups, bins of 8):", dict(sorted(hs.items())))
    if detail:
        for p in phases:
            print(f"{(p[1] - t0) / 1e3:10.1f} us  {p[0]}  span {(p[2] - p[1]) / 1e3:8.1f}  kernels {p[3] / 1e3:8.1f} ({p[5]:3d})  streams {p[4]:4d}  beside the other kind {p[6] / 1e3:8.1f}")


if __name__ == "__main__":
    main()

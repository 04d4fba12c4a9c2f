#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 rocpd (.db) kernel trace: per queue and for all queues together, the
fraction of the wall time (between the first and the last kernel of the LAST `frac` of the trace) during which at least
one kernel was running, and the share of that time with kernels of two queues running at once.
Usage: python tools/rocpd_busy.py <results.db> [frac=0.4 | <N>ms = the last N milliseconds of the trace]"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    con = sqlite3.connect(sys.argv[1])
    arg = sys.argv[2] if len(sys.argv) > 2 else "0.4"
    last_ns = float(arg[:-2]) * 1e6 if arg.endswith("ms") else None
    frac = 0.4 if last_ns else float(arg)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in kcols else "stream_id"
    rows = list(cur.execute(f"select start, end, {qcol} from {kd} order by start"))
    t_end = rows[-1][1]
    t_beg = t_end - last_ns if last_ns else rows[0][0] + (1.0 - frac) * (t_end - rows[0][0])
    rows = [r for r in rows if r[0] >= t_beg]
    wall = rows[-1][1] - rows[0][0]
    qs = sorted(set(r[2] for r in rows), key=lambda q: -sum(r[1] - r[0] for r in rows if r[2] == q))
    busy = {q: union([(r[0], r[1]) for r in rows if r[2] == q]) for q in qs}
    allb = union([(r[0], r[1]) for r in rows])
    print(f"window {wall / 1e6:.1f} ms, {len(rows)} kernels")
    for q in qs[:4]:
        print(f"queue {q}: busy {100.0 * busy[q] / wall:5.1f} %  ({sum(1 for r in rows if r[2] == q)} kernels)")
    print(f"any queue busy: {100.0 * allb / wall:5.1f} %   two queues at once: {100.0 * (sum(busy.values()) - allb) / wall:5.1f} % (sum of queue busy - union)")
    # idle gaps (no kernel of any queue running) by length
    iv = sorted((r[0], r[1]) for r in rows)
    gaps, ce = [], iv[0][1]
    for st, en in iv[1:]:
        if st > ce:
            gaps.append(st - ce)
        ce = max(ce, en)
    classes = [(0, 5e3), (5e3, 20e3), (20e3, 50e3), (50e3, 100e3), (100e3, 300e3), (300e3, 1e12)]
    print("idle gaps: " + ", ".join(
        f"{lo / 1e3:.0f}-{'inf' if hi > 1e11 else f'{hi / 1e3:.0f}'} us: {sum(1 for g in gaps if lo <= g < hi)} = "
        f"{100.0 * sum(g for g in gaps if lo <= g < hi) / wall:.1f} %" for lo, hi in classes))


if __name__ == "__main__":
    main()

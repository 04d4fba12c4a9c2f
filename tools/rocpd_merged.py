#!/usr/bin/env python3
"""Merged timeline of ALL queues from a rocprofv3 rocpd (.db) kernel trace: start (us, relative), duration, queue, how many
kernels of OTHER queues were running when this one started, workgroups, name - for a slice of the trace.
Usage: python tools/rocpd_merged.py <results.db> <start_ms_from_end> <length_ms>   (e.g. 100 4 = 4 ms starting 100 ms before the end)"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    back, length = float(sys.argv[2]), float(sys.argv[3])
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
    t0 = t_end - back * 1e6
    t1 = t0 + length * 1e6
    sel = [r for r in rows if r[1] >= t0 and r[1] < t1]
    print(f"{len(sel)} kernels in [{-back:.1f} ms, {-back + length:.1f} ms] from the end of the trace")
    for r in sel:
        others = sum(1 for o in rows if o[3] != r[3] and o[1] <= r[1] < o[2])
        name = r[0].replace("_Z21dec_layer_attn_kernelILi256ELi32ELi10E", "dec_attn<").replace("_Z16ffn_fused_kernelILi256E", "ffn<").replace("_Z23dec_layer_stream_kernelI", "dec_stream<")[:46]
        print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:7.2f}  q{r[3]}  other-queue kernels running at start: {others}  wgs {r[4]:5d}  {name}")


if __name__ == "__main__":
    main()

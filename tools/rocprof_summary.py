#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the text files kept under profiles/.

    python tools/rocprof_summary.py stats  <trace.db>            -> per-kernel calls / total / average (us)
    python tools/rocprof_summary.py pmc    <pmc.db> [COUNTER]    -> per-kernel average counter value per dispatch
"""
import sqlite3
import sys


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
    for name, calls, tot, avg, pct in rows:
        print(f"{calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {name[:110]}")


def pmc(db, counter=None):
    con = sqlite3.connect(db)
    q = "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection"
    if counter:
        q += f" where counter_name = '{counter}'"
    q += " group by kernel_name, counter_name order by avg(value) desc"
    print(f"{'n':>5} {'avg':>14} {'min':>14} {'max':>14}  counter / kernel")
    for k, c, n, a, lo, hi in con.execute(q):
        print(f"{n:5d} {a:14.3f} {lo:14.3f} {hi:14.3f}  {c}  {k[:90]}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)

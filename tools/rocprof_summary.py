#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the text files kept under profiles/.

    python tools/rocprof_summary.py stats  <trace.db>            -> per-kernel calls / total / average (us)
    python tools/rocprof_summary.py pmc    <pmc.db> [COUNTER]    -> per-kernel average counter value per dispatch
    python tools/rocprof_summary.py split  <trace.db> [PATTERN]  -> per-kernel AND per-launch-shape (grid x workgroup size) calls /
                                                                    total / average / min / max (us): a kernel that is launched on
                                                                    problems of different sizes (K1m: key-points and key-lines) gets
                                                                    one line per shape, so the dominant launch can be read off alone
    python tools/rocprof_summary.py timeline <trace.db> [N]      -> the last N dispatches in start order: start (us, relative), duration,
                                                                    queue, launch shape, kernel — which launches overlap on the two
                                                                    streams of a step, and which one the next stage waits for
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


def _table_like(con, stem):
    """rocpd names its tables `rocpd_<stem>_<uuid>` and offers views without the suffix; take whichever exists."""
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table', 'view')")]
    if stem in names:
        return stem
    for prefix in (f"rocpd_{stem}", stem):
        for n in names:
            if n.startswith(prefix):
                return n
    return None


def split(db, pattern=None):
    con = sqlite3.connect(db)
    view = _table_like(con, "kernels")   # the `kernels` view: name, start, end, grid_size*, workgroup_size*, ...
    if view is None:
        raise SystemExit("no kernel dispatch table in " + db)
    cols = [r[1] for r in con.execute(f"pragma table_info('{view}')")]

    def pick(*cands):
        for c in cands:
            if c in cols:
                return c
        return None
    name, start, end = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
    gx, wx = pick("grid_size_x", "grid_x", "grid_size"), pick("workgroup_size_x", "workgroup_x", "workgroup_size")
    if None in (name, start, end, gx, wx):
        raise SystemExit(f"unexpected columns in {view}: {cols}")
    q = (f"select {name}, {gx}, {wx}, count(*), sum({end} - {start}) / 1e3, avg({end} - {start}) / 1e3, min({end} - {start}) / 1e3, "
         f"max({end} - {start}) / 1e3 from {view}")
    if pattern:
        q += f" where {name} like '%{pattern}%'"
    q += f" group by {name}, {gx}, {wx} order by sum({end} - {start}) desc"
    print(f"{'calls':>6} {'grid_x':>10} {'wg_x':>5} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for k, g, w, n, tot, avg, lo, hi in con.execute(q):
        print(f"{n:6d} {g:10d} {w:5d} {tot:12.1f} {avg:10.2f} {lo:10.2f} {hi:10.2f}  {k[:100]}")


def timeline(db, n=60, skip=0):  # the n dispatches in front of the last `skip`, in start order; skip < 0: the n dispatches behind the first -skip
    con = sqlite3.connect(db)
    view = _table_like(con, "kernels")
    cols = [r[1] for r in con.execute(f"pragma table_info('{view}')")]

    def pick(*cands):
        for c in cands:
            if c in cols:
                return c
        return None
    name, start, end = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
    gx, wx = pick("grid_size_x", "grid_x", "grid_size"), pick("workgroup_size_x", "workgroup_x", "workgroup_size")
    queue = pick("queue_id", "queue", "stream_id", "stream") or "0"
    if int(skip) < 0:
        rows = con.execute(f"select {start}, {end}, {queue}, {gx}, {wx}, {name} from {view} order by {start} asc limit {int(n)} offset {-int(skip)}").fetchall()
    else:
        rows = con.execute(f"select {start}, {end}, {queue}, {gx}, {wx}, {name} from {view} order by {start} desc limit {int(n)} offset {int(skip)}").fetchall()[::-1]
    t0 = rows[0][0]
    print(f"{'start_us':>10} {'end_us':>10} {'dur_us':>9} {'queue':>6} {'grid_x':>9} {'wg':>5}  kernel")
    for st, en, q, g, w, k in rows:
        print(f"{(st - t0) / 1e3:10.1f} {(en - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} {str(q):>6} {g:9d} {w:5d}  {k[:90]}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 60, sys.argv[4] if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == "split":
        split(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)

"""Ordered kernel list of ONE step from a rocprofv3 rocpd .db: start offset, duration, queue, name.

    python scripts/rocpd_sequence.py <db> [attn launches per step = 12] [step = 4] [min_us = 0]
"""
import sqlite3
import sys


def main(path, per_step=12, step=4, min_us=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
    rows = cur.execute(sel).fetchall()
    marks = [s for (n, s, e, q) in rows if "vit_attn_kernel" in n][::per_step]
    t0, t1 = marks[step], marks[step + 1]
    # the step's own boundary lies before the first attention launch: back up to the previous gap-free start
    queues = {}
    print("# offset_us,dur_us,queue,name")
    for n, s, e, q in rows:
        if s < t0 or s >= t1:
            continue
        qi = queues.setdefault(q, len(queues))
        d = (e - s) / 1e3
        if d >= min_us:
            print(f"{(s - t0) / 1e3:10.1f},{d:9.1f},{qi},{n[:110]}")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 12, int(a[3]) if len(a) > 3 else 4, float(a[4]) if len(a) > 4 else 0.0)

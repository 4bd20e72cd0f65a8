"""Critical-path composition of the step from a rocprofv3 rocpd .db (kernel trace).

For a window of whole steps (from the first ViT attention launch of step `skip` to that of the last step) every
instant is attributed: idle (no kernel resident), solo (exactly one kernel resident: its time is on the critical
path) or shared (several resident: split equally).  Prints per-kernel solo / shared / total ms per step.

    python scripts/rocpd_timeline.py <db> [attn launches per step = 12] [steps to skip = 3] [marker kernel = vit_attn_kernel]
"""
import collections
import sqlite3
import sys


def main(path, per_step=12, skip=3, marker="vit_attn_kernel"):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [s for (n, s, e) in rows if marker in n][::per_step]
    t0, t1, steps = marks[skip], marks[-1], len(marks) - 1 - skip
    ev = []
    for i, (n, s, e) in enumerate(rows):
        s, e = max(s, t0), min(e, t1)
        if e > s:
            ev.append((s, 1, i))
            ev.append((e, 0, i))
    ev.sort()
    live = set()
    solo, shared, total = collections.Counter(), collections.Counter(), collections.Counter()
    idle = 0
    last = t0
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            if not live:
                idle += dt
            elif len(live) == 1:
                solo[rows[next(iter(live))][0]] += dt
            else:
                for j in live:
                    shared[rows[j][0]] += dt / len(live)
        last = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    for n, s, e in rows:
        s, e = max(s, t0), min(e, t1)
        if e > s:
            total[n] += e - s
    span = (t1 - t0) / steps / 1e6
    print(f"# {steps} steps, {span:.2f} ms per step; idle {idle / steps / 1e6:.2f} ms per step")
    print("name,solo_ms,shared_ms,total_ms   (per step)")
    names = sorted(total, key=lambda n: -(solo[n] + shared[n]))
    for n in names[:45]:
        print(f"\"{n[:90]}\",{solo[n] / steps / 1e6:.3f},{shared[n] / steps / 1e6:.3f},{total[n] / steps / 1e6:.3f}")
    rest = names[45:]
    print(f"\"(other {len(rest)} kernels)\",{sum(solo[n] for n in rest) / steps / 1e6:.3f},"
          f"{sum(shared[n] for n in rest) / steps / 1e6:.3f},{sum(total[n] for n in rest) / steps / 1e6:.3f}")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 12, int(a[3]) if len(a) > 3 else 3, a[4] if len(a) > 4 else "vit_attn_kernel")

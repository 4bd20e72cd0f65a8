"""Summarise a rocprofv3 rocpd .db (kernel trace) as a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {total/1e6:.3f} ms total kernel time")
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for r in rows[:top]:
        print(f"\"{r[0][:140]}\",{r[1]},{r[2]},{r[3]:.0f},{100*r[2]/total:.2f},{r[4]},{r[5]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)

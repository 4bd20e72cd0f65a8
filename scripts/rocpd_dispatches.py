"""Per-dispatch durations of the kernels whose name contains a pattern, in launch order, from a rocprofv3 rocpd .db (kernel trace):
python scripts/rocpd_dispatches.py <db> <pattern> [first N]  ->  index, start offset (us), duration (us), grid, name"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
sel = f"select {name_col}, start, end" + (f", {gx}" if gx else "") + f" from kernels where {name_col} like ? order by start"
rows = cur.execute(sel, (f"%{sys.argv[2]}%",)).fetchall()
t0 = rows[0][1] if rows else 0
for i, r in enumerate(rows[: int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)]):
    print(f"{i:4d} {(r[1] - t0) / 1e3:12.1f} {(r[2] - r[1]) / 1e3:9.1f} us  grid {r[3] if gx else '?'}  {r[0][:70]}")

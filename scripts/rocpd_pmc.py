"""Per-kernel mean of a PMC counter from a rocprofv3 rocpd .db (values in the counter's own unit).
    rocpd_pmc.py <db> [pattern]              one row per (kernel, counter): dispatches, mean value, mean duration
    rocpd_pmc.py <db> <pattern> --dispatches one row per dispatch in launch order (for kernels whose launches differ in shape)"""
import sqlite3
import sys


def main(path, pattern="unopose", mode=""):
    db = sqlite3.connect(path)
    if mode == "--dispatches":
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")] or \
               [d[0] for d in db.execute("select * from counters_collection limit 0").description]
        order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
        rows = db.execute(f"select kernel_name, counter_name, sum(value), max(duration), {order} from counters_collection "
                          f"where kernel_name like ? group by {order}, kernel_name, counter_name order by {order}", (f"%{pattern}%",)).fetchall()
        print("Kernel,Counter,Index,Value,DurationNs")
        for i, r in enumerate(rows):
            print(f"\"{r[0][:110]}\",{r[1]},{i},{r[2]:.3f},{r[3]:.0f}")
        return
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                      "where kernel_name like ? group by kernel_name, counter_name order by 4 desc",
                      (f"%{pattern}%",)).fetchall()
    print("Kernel,Counter,Dispatches,MeanValue,MeanDurationNs")
    for r in rows:
        print(f"\"{r[0][:110]}\",{r[1]},{r[2]},{r[3]:.3f},{r[4]:.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:])

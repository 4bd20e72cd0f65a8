"""Per-kernel mean of a PMC counter from a rocprofv3 rocpd .db (values in the counter's own unit)."""
import sqlite3
import sys


def main(path, pattern="unopose"):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                      "where kernel_name like ? group by kernel_name, counter_name order by 4 desc",
                      (f"%{pattern}%",)).fetchall()
    print("Kernel,Counter,Dispatches,MeanValue,MeanDurationNs")
    for r in rows:
        print(f"\"{r[0][:110]}\",{r[1]},{r[2]},{r[3]:.3f},{r[4]:.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:])

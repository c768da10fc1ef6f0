"""Per-dispatch counter values of the kernels whose name contains `pattern`, in dispatch order, from rocprofv3 rocpd databases
(one --pmc pass each).  usage: pmc_launches.py pattern db1 [db2 ...]"""
import sqlite3, sys
from collections import defaultdict

pat = sys.argv[1]
rows = defaultdict(dict)
order = {}
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
    idc = "dispatch_id" if "dispatch_id" in cols else ("event_id" if "event_id" in cols else None)
    q = f"select name, counter_name, counter_value, duration, start{', ' + idc if idc else ''} from pmc_events"
    # (one row per dispatch, counter and hardware instance: a dispatch = one `start` stamp; instances are summed)
    starts = {}
    for r in sorted(con.execute(q), key=lambda r: r[4]):
        name, cname, val, dur, start = r[:5]
        if pat not in name:
            continue
        k = starts.setdefault(start, len(starts))
        rows[k][cname] = rows[k].get(cname, 0) + val
        rows[k].setdefault("us", dur / 1e3)
cn = sorted({c for r in rows.values() for c in r if c != "us"})
print("idx | us | " + " | ".join(cn))
for k in sorted(rows):
    print(k, "| %.0f | " % rows[k]["us"] + " | ".join("%.4g" % rows[k].get(c, float("nan")) for c in cn))

"""Per-kernel PMC averages from rocprofv3 rocpd databases (one --pmc pass per database).
usage: rocpd_pmc.py out.md db1 db2 ..."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"_ZN2lk\d+([a-z0-9_]+?)(?:I(.*))?E[vPK]", name)
    if m:  # mangled (templated on a config struct, or a plain lk:: kernel): readable name + integer template arguments
        ints = re.findall(r"Li(\d+)E", m.group(2) or "")
        b = re.findall(r"Lb([01])E", m.group(2) or "")
        name = "lk::" + m.group(1) + ("<" + ",".join(ints + b) + ">" if (ints or b) else "")
    return name if len(name) < 90 else name[:87] + "..."


def main(out, dbs):
    table = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
    dur = defaultdict(list)
    for db in dbs:
        con = sqlite3.connect(db)
        for name, cname, val, d in con.execute("select name, counter_name, counter_value, duration from pmc_events"):
            if not name.startswith(("lk::", "void lk::", "_ZN2lk")):
                continue
            table[short(name)][cname].append(val)
            dur[short(name)].append(d)
    counters = sorted({c for k in table for c in table[k]})
    lines = ["| kernel | launches | avg us | " + " | ".join(counters) + " |", "|---|---|---|" + "---|" * len(counters)]
    for k in sorted(table):
        n = max(len(v) for v in table[k].values())
        row = [f"`{k}`", str(n), f"{sum(dur[k]) / len(dur[k]) / 1e3:.1f}"]
        for c in counters:
            v = table[k].get(c, [])
            row.append(f"{sum(v) / len(v):.4g}" if v else "")
        lines.append("| " + " | ".join(row) + " |")
    text = "\n".join(lines)
    print(text)
    open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

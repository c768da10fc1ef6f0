"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (like
`--stats` csv): calls, total / average / min / max duration, share of GPU kernel time."""
import sqlite3
import sys


def main(db, out=None, top=40):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels"))
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.2f} ms over {len(rows)} dispatches")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

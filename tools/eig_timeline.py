"""Timeline of ONE `decompose` (43 factors) from a rocprofv3 --kernel-trace rocpd database of `eig_profile.py decompose`:
per queue the busy share, the union busy share, mean gap between consecutive kernels of a queue, per-kernel durations.
usage: eig_timeline.py <db> [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else "0"
    scol = "stream_id" if "stream_id" in cols else "0"
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    rows = sorted(cur.execute(f"select name, start, end, {qcol}, {scol}, {gcol} from kernels"), key=lambda r: r[1])
    lines = ["columns: " + ", ".join(cols)]
    gathers = [r for r in rows if "eig_gather" in r[0]]
    ndec = 5  # eig_profile.py runs five decompositions (n_streams 3, 2, 3, 2, 3): the window is the last one
    per = len(gathers) // ndec
    t0 = max(r[2] for r in gathers[:(ndec - 1) * per])
    t1 = max(r[2] for r in gathers)
    rows = [r for r in rows if r[1] >= t0 and r[2] <= t1 and "eig" in r[0]]
    t0 = rows[0][1]
    span = t1 - t0
    lines.append(f"window {span / 1e6:.1f} ms, {len(rows)} dispatches, sum of durations {sum(r[2] - r[1] for r in rows) / 1e6:.1f} ms")
    ev = []
    for r in rows:
        ev.append((r[1], 1)), ev.append((r[2], -1))
    ev.sort()
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    lines.append("kernels in flight -> share of the window: " + ", ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(hist.items())))
    byq = {}
    for r in rows:
        byq.setdefault((r[3], r[4]), []).append(r)
    for q, rs in sorted(byq.items()):
        b = sum(r[2] - r[1] for r in rs)
        gaps = [b2[1] - a[2] for a, b2 in zip(rs, rs[1:])]
        small = [g for g in gaps if g < 100e3]
        lines.append(f"queue/stream {q}: {len(rs)} dispatches, busy {100 * b / span:.1f} %, first {1e-6 * (rs[0][1] - t0):.1f} ms, last {1e-6 * (rs[-1][2] - t0):.1f} ms, "
                     f"mean gap {sum(small) / max(1, len(small)) / 1e3:.1f} us (gaps < 100 us), gap total {sum(gaps) / 1e6:.1f} ms")
    agg = {}
    for r in rows:
        key = (r[0].split("(")[0][-40:], r[5])
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    for (name, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        lines.append(f"  {a[1] / 1e6:7.1f} ms  {a[0]:6d} x {a[1] / a[0] / 1e3:7.1f} us  grid {g}  {name}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

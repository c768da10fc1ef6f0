"""Timeline of the steady-state c4 step from a rocprofv3 --kernel-trace rocpd database: per hardware queue the busy time,
the union busy time (any kernel running), the time with two or more kernels in flight, and the largest idle gaps — to see
whether the step is bound by one stream's chain, by the sum of the work, or by gaps.  Development tool.
usage: timeline.py <db> [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = sorted(cur.execute(f"select {sel} from kernels"), key=lambda r: r[1])
    # steady state: eight whole steps, cut at a kernel that runs once per step (the likelihood's Hessian root)
    marks = [r[1] for r in rows if "softmax_hess" in r[0]]
    nsteps = min(8, len(marks) - 2)
    t0, t1 = marks[-1 - nsteps], marks[-1]
    rows = [r for r in rows if r[1] >= t0 and r[1] < t1]
    span = t1 - t0
    lines = [f"window {span / 1e6:.2f} ms = {nsteps} steps of {span / 1e6 / nsteps:.2f} ms, {len(rows)} dispatches"]
    # union / overlap by sweep line
    ev = []
    for r in rows:
        ev.append((r[1], 1)), ev.append((r[2], -1))
    ev.sort()
    depth, last, busy, multi = 0, t0, 0, 0
    gaps = []
    ev = [(min(t, t1), d) for t, d in ev]
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        elif t - last > 0:
            gaps.append((t - last, last))
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    lines.append(f"any kernel running: {100 * busy / span:.1f} %   two or more: {100 * multi / span:.1f} %   idle: {100 * (span - busy) / span:.1f} %")
    lines.append(f"sum of kernel durations / window: {sum(r[2] - r[1] for r in rows) / span:.3f}")
    byq = {}
    for r in rows:
        byq.setdefault((r[3], r[4]), []).append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        b = sum(r[2] - r[1] for r in rs)
        lines.append(f"queue/stream {q}: {len(rs)} dispatches, busy {100 * b / span:.1f} % of the window")
    gaps.sort(reverse=True)
    lines.append("largest idle gaps (us): " + ", ".join(f"{g / 1e3:.0f}" for g, _ in gaps[:12]) + f"   total idle in gaps > 5 us: {sum(g for g, _ in gaps if g > 5e3) / 1e6:.2f} ms")
    # duration inflation: per kernel name, mean duration when running alone vs overlapped is not separable here; list top kernels
    agg = {}
    for r in rows:
        a = agg.setdefault(r[0][:70], [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        lines.append(f"  {a[1] / span * 100:5.1f} %  {a[0]:5d} x {a[1] / a[0] / 1e3:7.1f} us  {name}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

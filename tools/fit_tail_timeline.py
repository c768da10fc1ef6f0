"""From a rocprofv3 --kernel-trace rocpd database of tools/fit_tail.py: the kernels of the LAST fit before its first and
after its last minibatch (marks: the likelihood's Hessian root, once per minibatch), in time order.  usage: <db> [out]"""
import sqlite3, sys

def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = sorted(cur.execute(f"select name, start, end, {scol} from kernels"), key=lambda r: r[1])
    marks = [i for i, r in enumerate(rows) if "softmax_hess" in r[0]]
    K = len(marks) // 3
    first, last = marks[-K], marks[-1]
    prev_last = marks[-K - 1]
    lines = []
    # steady state of the last fit: mark to mark
    span = (rows[last][1] - rows[first][1]) / (K - 1)
    lines.append(f"K = {K}; steady state {span / 1e6:.3f} ms per step (mark to mark)")
    # tail: from the last mark + one steady step to the last kernel
    t_tail0 = rows[last][1]
    t_end = max(r[2] for r in rows[last:])
    lines.append(f"after the last mark: {(t_end - t_tail0) / 1e6:.2f} ms until the last kernel ends (a steady step is {span / 1e6:.2f})")
    # head: previous fit's last kernel end -> this fit's first mark
    prev_end = max(r[2] for r in rows[prev_last:first] if r[1] < rows[first][1] - 1.5 * span) if first > prev_last else rows[first][1]
    lines.append("---- tail kernels (start offset us from the last mark, duration us, stream, name)")
    for r in rows[last:]:
        if r[2] - r[1] > 15e3 or "assemble" in r[0] or "symmetr" in r[0] or "permute" in r[0]:
            lines.append(f"{(r[1] - t_tail0) / 1e3:9.0f} {(r[2] - r[1]) / 1e3:8.1f}  s{r[3]}  {r[0][:90]}")
    agg = {}
    for r in rows[last:]:
        if r[1] > t_tail0 + span:
            a = agg.setdefault(r[0][:80], [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
    lines.append("---- kernels starting later than one steady step after the last mark, by name")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"{a[1]:9.1f} us {a[0]:5d} x  {n}")
    # head
    h = [r for r in rows[prev_last:first + 1] if r[1] > rows[prev_last][1] + 3 * span]
    if h:
        lines.append(f"---- head: {len(h)} kernels from {(rows[first][1] - h[0][1]) / 1e3:.0f} us before the first mark")
        agg = {}
        for r in h:
            a = agg.setdefault(r[0][:80], [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            lines.append(f"{a[1]:9.1f} us {a[0]:5d} x  {n}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")

main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

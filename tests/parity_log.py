"""Worst relative error each test measured through its ``rel`` helper (test infrastructure): ``record_error`` is called by
the helpers of the -m gpu tests, ``tests/conftest.py`` writes the table to gpurun_out/parity_errors.log when the session
ends — the file the round's `profiles/rNN_parity_errors.log` is a copy of."""
import os

PARITY = {}  # test id -> [comparisons, worst relative error]


def record_error(value: float, tag: str | None = None) -> float:
    """``tag``: a separate log entry of the same test (e.g. what the fp32 CPU oracle itself measures against fp64, kept
    apart from the kernels' own error)"""
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + (f" <{tag}>" if tag else "")
    ent = PARITY.setdefault(tid, [0, 0.0])
    ent[0] += 1
    if value == value:
        ent[1] = max(ent[1], float(value)) if ent[1] == ent[1] else ent[1]
    else:  # NaN
        ent[1] = float("nan")
    return value

"""Ordered kernel timeline of ONE training step (the last timed one) from a rocprofv3 --kernel-trace CSV of bench.py:
start offset, duration, gap to the previous kernel on the same queue, short kernel name.  Complements
tools/prof_window.py (aggregates): shows which small launches sit next to each other and where the stream idles.

Usage: python tools/prof_timeline.py <kernel_trace.csv> [min_us_to_print]
"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:110]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path, newline="")))
    name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k], r.get("Queue_Id", "0")) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam_flat_kernel" in e[2]]
    if len(adam) < 2:
        raise SystemExit(f"prof_timeline: found {len(adam)} `adam_flat_kernel` launches, need two to delimit a step")
    i0, i1 = adam[-2] + 1, adam[-1] + 1
    step = ev[i0:i1]
    t0 = step[0][0]
    last_end = {}
    print(f"# last step: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us wall")
    print("# start_us  dur_us  gap_us  queue  kernel")
    for s, e, n, q in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(last_end.get(q, 0), e)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {q:>3s}  {short(n)}")


if __name__ == "__main__":
    main()

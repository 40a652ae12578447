"""Per-kernel table of the launches between the first two `delay_kernel` markers of a rocprofv3 --kernel-trace CSV
(tools/eval_profile_run.py brackets N inference forwards with them).
Usage: python tools/prof_between.py <kernel_trace.csv> <N forwards>"""
import collections
import sys

from prof_window import func_name, load

MARK = "delay_kernel"


def main():
    ev = load(sys.argv[1])
    n = int(sys.argv[2])
    marks = [e for e in ev if MARK in e[2]]
    if len(marks) < 2:
        raise SystemExit(f"prof_between: found {len(marks)} `{MARK}` markers, need 2")
    t0, t1 = marks[0][1], marks[1][0]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    if not win:
        raise SystemExit("prof_between: empty window")
    agg = collections.OrderedDict()
    for s, e, name, q in win:
        a = agg.setdefault(func_name(name), [0.0, 0])
        a[0] += (e - s) * 1e-6
        a[1] += 1
    busy = sum(a[0] for a in agg.values())
    lib_ms = sum(a[0] for k, a in agg.items() if k.startswith("Cijk_") or "miopen" in k.lower() or "rocblas" in k.lower())
    nat_ms = sum(a[0] for k, a in agg.items() if k.startswith("at::native"))
    print(f"window: {n} forwards, {(t1 - t0) * 1e-6 / n:.3f} ms/forward wall, {busy / n:.3f} ms/forward summed kernel time in "
          f"{len(win) / n:.0f} kernels/forward")
    print(f"by origin: rocBLAS/MIOpen {lib_ms / n:.3f} ms/forward, at::native {nat_ms / n:.3f} ms/forward, own kernels "
          f"{(busy - lib_ms - nat_ms) / n:.3f} ms/forward")
    print("ms/forward,calls/forward,avg_us,function")
    for k, (ms, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{ms / n:.4f},{c / n:.1f},{ms / c * 1e3:.1f},\"{k}\"")


if __name__ == "__main__":
    main()

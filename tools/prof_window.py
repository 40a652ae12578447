"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 --kernel-trace CSV of `bench.py`.

The whole-run `--stats` summary also contains the warm-up steps and the op micro-benchmarks; `adam_flat_kernel` (one launch
per training step, the step's last kernel) marks the end of every step, so the window
[end of step (total - K) ... end of the last step] holds exactly the K timed steps.  The queue that carries
`adam_flat_kernel` is the step's own stream ("main"; the queues of the step's forked branches count with it); the queue that
carries the sampling kernels of the next batch's pyramid is the side stream.  The table is ordered by summed MAIN-stream time: its first row is the kernel `bench.py` reports as `roofline`.

Usage: python tools/prof_window.py <kernel_trace.csv> <K timed steps> [top N]
Exits non-zero (and prints nothing to stdout) when the marker kernel is missing or there are fewer than K + 1 steps.
"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
MARKER = "adam_flat_kernel"


def func_name(n):
    """kernel function without template arguments / parameter list: the key bench.py matches on."""
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.split(r"[<(]", n, 1)[0].strip()


def load(path):
    rows = list(csv.DictReader(open(path, newline="")))
    if not rows:
        raise SystemExit(f"prof_window: {path} holds no kernel rows")
    name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k], r.get("Queue_Id", "0")) for r in rows)
    return ev


def window(ev, K):
    marks = [e for e in ev if MARKER in e[2]]
    if len(marks) < K + 1:
        raise SystemExit(f"prof_window: found {len(marks)} `{MARKER}` launches, need at least {K + 1} "
                         f"(one per step marks the step's end; was the optimizer kernel renamed or the trace cut short?)")
    t0, t1 = marks[-K - 1][1], marks[-1][1]
    main_q = marks[-1][3]
    win = [e for e in ev if t0 <= e[0] <= t1]
    # Since round 6 the captured step forks: the relation head runs beside the caption decoder and the captioner's weight
    # gradients beside the detector's backward, on queues of their own.  Those launches belong to the STEP; the side stream is
    # the queue that carries the next batch's sampling chain (its first-level kernel marks it).
    side_qs = {e[3] for e in win if "fps_bucket_kernel" in e[2] or "delay_kernel" in e[2]} - {main_q}
    if side_qs:
        win = [(s, e, n, (q if q in side_qs else main_q)) for s, e, n, q in win]
    return win, t0, t1, main_q


def main():
    path, K = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    win, t0, t1, main_q = window(load(path), K)
    tot, cnt, side = collections.Counter(), collections.Counter(), collections.Counter()
    for s, e, n, q in win:
        key = n
        (tot if q == main_q else side)[key] += e - s
        cnt[key] += 1
    wall = (t1 - t0) / 1e6 / K
    main_busy = sum(tot.values()) / 1e6 / K
    side_busy = sum(side.values()) / 1e6 / K
    n_main = sum(1 for e in win if e[3] == main_q)
    print(f"timed window: {K} steps, {wall:.3f} ms/step wall, main stream {main_busy:.3f} ms/step summed kernel time in "
          f"{n_main / K:.0f} kernels/step, side stream(s) {side_busy:.3f} ms/step in {(len(win) - n_main) / K:.0f} kernels/step")
    main = sorted(e for e in win if e[3] == main_q)
    idle, end, gaps = 0, main[0][1], collections.Counter()
    for s, e, n, q in main[1:]:
        if s > end:
            idle += s - end
            gaps[min((s - end) // 1000, 20)] += 1
        end = max(end, e)
    print(f"main stream: {idle / 1e6 / K:.3f} ms/step idle between kernels; gap histogram (us: count/step) "
          + ", ".join(f"{k}{'+' if k == 20 else ''}: {v / K:.0f}" for k, v in sorted(gaps.items())))
    # the same kernels grouped by function (templates merged): what `roofline` names
    by_fn, by_fn_cnt = collections.Counter(), collections.Counter()
    for n, v in tot.items():
        by_fn[func_name(n)] += v
    for s, e, n, q in main:
        by_fn_cnt[func_name(n)] += 1
    lib = sum(v for n, v in by_fn.items() if n.startswith("Cijk_") or "naive_conv" in n or "miopen" in n.lower())
    torch_glue = sum(v for n, v in by_fn.items() if n.startswith("at::native") or n.startswith("at::cuda"))
    print(f"main stream by origin: rocBLAS/MIOpen {lib / 1e6 / K:.3f} ms/step, at::native {torch_glue / 1e6 / K:.3f} ms/step, "
          f"own kernels {(sum(by_fn.values()) - lib - torch_glue) / 1e6 / K:.3f} ms/step")
    print("main_ms/step,calls/step,avg_us,function")
    for n, v in by_fn.most_common(top):
        print(f"{v / 1e6 / K:.4f},{by_fn_cnt[n] / K:.1f},{v / by_fn_cnt[n] / 1e3:.1f},\"{n[:150]}\"")
    print("# per instantiation (main stream first, then side stream)")
    print("ms/step,calls/step,avg_us,stream,kernel")
    for n, v in tot.most_common(top):
        print(f"{v / 1e6 / K:.4f},{cnt[n] / K:.1f},{v / cnt[n] / 1e3:.1f},main,\"{n[:200]}\"")
    for n, v in side.most_common(top):
        print(f"{v / 1e6 / K:.4f},{cnt[n] / K:.1f},{v / cnt[n] / 1e3:.1f},side,\"{n[:200]}\"")


if __name__ == "__main__":
    main()

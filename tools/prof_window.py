"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 --kernel-trace CSV of `bench.py`.

The whole-run `--stats` summary also contains MIOpen's solver search and the op micro-benchmarks; the fused Adam
kernel marks the end of every training step, so the window [end of step (total - K) ... end of the last step]
holds exactly the K timed steps.
Usage: python tools/prof_window.py <kernel_trace.csv> <K timed steps> [top N]
"""
import collections
import csv
import sys


def main():
    path, K = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.DictReader(open(path)))
    name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k]) for r in rows]
    ev.sort()
    adam = [e for e in ev if "FusedOptimizer" in e[2] or "multi_tensor_apply" in e[2]]
    # group Adam kernels into steps (gaps > 1 ms separate steps)
    steps, cur = [], [adam[0]]
    for e in adam[1:]:
        if e[0] - cur[-1][1] > 1_000_000:
            steps.append(cur)
            cur = [e]
        else:
            cur.append(e)
    steps.append(cur)
    t0 = steps[-K - 1][-1][1]
    t1 = steps[-1][-1][1]
    win = [e for e in ev if t0 <= e[0] <= t1]
    tot = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in win:
        tot[n] += e - s
        cnt[n] += 1
    wall = (t1 - t0) / 1e6 / K
    busy = sum(tot.values()) / 1e6 / K
    print(f"timed window: {K} steps, {wall:.2f} ms/step wall, {busy:.2f} ms/step summed kernel time, "
          f"{len(win) / K:.0f} kernels/step")
    # idle time of the step's own stream: gaps between consecutive kernels (the side-stream sampling pyramid excluded)
    main = sorted(e for e in win if "fps_" not in e[2])
    idle, end, gaps = 0, main[0][1], collections.Counter()
    for s, e, n in main[1:]:
        if s > end:
            idle += s - end
            gaps[min((s - end) // 1000, 20)] += 1
        end = max(end, e)
    print(f"main stream: {idle / 1e6 / K:.2f} ms/step idle between kernels; gap histogram (us: count/step) "
          + ", ".join(f"{k}{'+' if k == 20 else ''}: {v / K:.0f}" for k, v in sorted(gaps.items())))
    big = collections.Counter()
    end, prev = main[0][1], main[0][2]
    for s_, e, n in main[1:]:
        if s_ - end > 20_000:
            big[(prev[:60], n[:60])] += s_ - end
        if e > end:
            end, prev = e, n
    for (a, b), v in big.most_common(12):
        print(f"   idle {v / 1e3 / K:7.1f} us/step between  {a}  ->  {b}")
    print("ms/step,calls/step,avg_us,kernel")
    for n, v in tot.most_common(top):
        print(f"{v / 1e6 / K:.3f},{cnt[n] / K:.1f},{v / cnt[n] / 1e3:.1f},\"{n[:150]}\"")


if __name__ == "__main__":
    main()

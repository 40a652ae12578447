"""Per launch position of the training step: mean duration in steps with the sampling chain beside them against steps without
(see side_cost.py).  Usage: python tools/lab/side_cost_diff.py <kernel_trace.csv> [top N]"""
import collections, csv, re, sys
import os
csv.field_size_limit(1 << 30)
MARK = os.environ.get("MARK", "fps_bucket")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:100]
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[nk], r.get("Queue_Id", "0")) for r in rows)
marks = [i for i, e in enumerate(ev) if "adam_flat_kernel" in e[2]]
# the side stream is the queue that carries the sampling chain; every other queue belongs to the step (its forked branches included)
side_qs = {e[3] for e in ev[marks[len(marks) // 2]:] if "fps_bucket_kernel" in e[2] or "delay_kernel" in e[2]}
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    t0, t1 = ev[a][1], ev[b][1]
    ks = [e for e in ev[a + 1:b + 1]] + [e for e in ev[max(0, a - 3):a + 1] if e[3] in side_qs and e[1] > ev[a][1]]
    main = [e for e in ks if e[3] not in side_qs]
    side = [e for e in ks if e[3] in side_qs]
    has_fps = any(MARK in e[2] for e in side)
    steps.append((has_fps, main, side, t1 - t0))
# keep graph-replayed steps only: the modal main-stream kernel count
cnt = collections.Counter(len(s[1]) for s in steps).most_common(1)[0][0]
A = [s for s in steps if s[0] and len(s[1]) == cnt][2:]
B = [s for s in steps if not s[0] and len(s[1]) == cnt and not s[2]][1:]
print(f"# {len(A)} steps with the chain, {len(B)} without; {cnt} main-stream kernels per step")
def mean_main(S): return sum(sum(e[1] - e[0] for e in s[1]) for s in S) / len(S) / 1e3
print(f"# main-stream kernel time: with {mean_main(A):.1f} us, without {mean_main(B):.1f} us, difference {mean_main(A) - mean_main(B):.1f} us")
pos = []
for i in range(cnt):
    a = sum(s[1][i][1] - s[1][i][0] for s in A) / len(A) / 1e3
    b = sum(s[1][i][1] - s[1][i][0] for s in B) / len(B) / 1e3
    off = sum(s[1][i][0] - s[1][0][0] for s in A) / len(A) / 1e3
    pos.append((a - b, a, b, off, i, short(A[0][1][i][2])))
print("# by launch position: diff_us with_us without_us start_offset_us(with) index kernel")
for d, a, b, off, i, n in sorted(pos, reverse=True)[:top]:
    print(f"{d:8.1f} {a:8.1f} {b:8.1f} {off:9.1f} {i:4d}  {n}")
by = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d, a, b, off, i, n in pos:
    k = re.split(r"[<(]", n, 1)[0]
    by[k][0] += a; by[k][1] += b; by[k][2] += 1
print("# by function: diff_us with_us without_us launches")
for k, (a, b, c) in sorted(by.items(), key=lambda kv: kv[1][1] - kv[1][0])[:40]:
    print(f"{a - b:8.1f} {a:8.1f} {b:8.1f} {c:4d}  {k}")
# in the chain's shadow or not: the step's time offset where the side stream ends
end_side = sum(max(e[1] for e in s[2]) - s[1][0][0] for s in A) / len(A) / 1e3
early = sum(p[0] for p in pos if p[3] < end_side); late = sum(p[0] for p in pos if p[3] >= end_side)
print(f"# side stream ends {end_side:.0f} us after the step's first kernel; extra main time before that {early:.1f} us, after {late:.1f} us")

# gaps on the main stream (start of kernel i - end of kernel i - 1), per position
def gaps(S, i): return sum(s[1][i][0] - s[1][i - 1][1] for s in S) / len(S) / 1e3
ga = [gaps(A, i) for i in range(1, cnt)]; gb = [gaps(B, i) for i in range(1, cnt)]
wa = sum(s[1][-1][1] - s[1][0][0] for s in A) / len(A) / 1e3; wb = sum(s[1][-1][1] - s[1][0][0] for s in B) / len(B) / 1e3
print(f"# first kernel start -> last kernel end: with {wa:.1f} us, without {wb:.1f} us; summed gaps with {sum(ga):.1f} us, without {sum(gb):.1f} us")
print("# largest gap increases: diff_us with_us without_us index kernel-after-gap")
for d, i in sorted(((ga[i] - gb[i], i) for i in range(cnt - 1)), reverse=True)[:25]:
    print(f"{d:8.2f} {ga[i]:8.2f} {gb[i]:8.2f} {i + 1:4d}  {short(A[0][1][i + 1][2])}   (after {short(A[0][1][i][2])[:40]})")
import statistics
print(f"# median gap with {statistics.median(ga):.2f} us, without {statistics.median(gb):.2f} us")

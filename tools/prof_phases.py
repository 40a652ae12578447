"""Per-phase kernel time of one training step (which module the GPU time goes to, forward and backward).

A one-element int16 fill is launched at every module boundary (forward pre-hook, backward pre-hook): its kernel
(`FillFunctor<short>`) is unique in the trace, so the kernels between two markers belong to the phase the first
marker opened.  The host records the order of the phase ids; the report matches them with the marker kernels.

  on the GPU box:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ph -- python3 tools/prof_phases.py run
    python tools/prof_phases.py report gpurun_out/ph/*/*kernel_trace.csv gpurun_out/phases_seq.json [top]
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import spacap3d_amd  # noqa: F401
    from spacap3d_amd.engine import Trainer, synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    from spacap3d_amd import synthetic as S

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_default().to(dev)
    tr = Trainer(model, S.mean_size_arr().numpy())
    batch = synthetic_batch(8, 40000, dev, seed=0)
    mk = torch.zeros(1, dtype=torch.int16, device=dev)
    seq = []

    def mark(name):
        seq.append(name)
        mk.fill_(1)

    bb = model.backbone_net
    mods = {"sa1": bb.sa1, "sa2": bb.sa2, "sa3": bb.sa3, "sa4": bb.sa4, "fp1": bb.fp1, "fp2": bb.fp2,
            "vote": model.vgen, "proposal": model.proposal, "caption": model.caption}
    cap = model.caption
    for n, m in cap.named_children():
        if sum(1 for _ in m.parameters()) > 0:
            mods["cap." + n] = m
    mods.pop("caption")
    for name, m in mods.items():
        m.register_forward_pre_hook(lambda mod, inp, name=name: mark("F:" + name))
        m.register_forward_hook(lambda mod, inp, out, name=name: mark("F:after_" + name))
        m.register_full_backward_pre_hook(lambda mod, g, name=name: mark("B:" + name))
    steps = 6
    for it in range(steps):
        mark("step_begin")
        loss = tr.step(batch)
        if it == 0:
            mark("setup_done")
    mark("end")
    torch.cuda.synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"seq": seq, "steps": steps}, open("gpurun_out/phases_seq.json", "w"))
    print("loss", float(loss), "markers", len(seq))


def report():
    path, seqp = sys.argv[2], sys.argv[3]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    meta = json.load(open(seqp))
    seq = meta["seq"]
    rows = list(csv.DictReader(open(path)))
    name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k]) for r in rows)
    ev = [e for e in ev if "fps_" not in e[2]]  # the sampling pyramid may sit on the side stream
    is_mark = lambda n: "FillFunctor<short>" in n
    nm = sum(1 for e in ev if is_mark(e[2]))
    if nm == len(seq) + 1:  # torch.zeros() of the marker tensor itself
        first_mark = next(i for i, e in enumerate(ev) if is_mark(e[2]))
        del ev[first_mark]
        nm -= 1
    assert nm == len(seq), (nm, len(seq))
    # last 3 steps only
    begins = [i for i, s in enumerate(seq) if s == "step_begin"]
    first = begins[-3]
    tot = collections.Counter()
    kern = collections.defaultdict(collections.Counter)
    kcnt = collections.defaultdict(collections.Counter)
    cnt = collections.Counter()
    mi, phase = -1, None
    big = []
    seqk = []
    for s, e, n in ev:
        if is_mark(n):
            mi += 1
            phase = seq[mi]
            continue
        if mi < first or phase is None:
            continue
        tot[phase] += e - s
        cnt[phase] += 1
        kern[phase][n] += e - s
        kcnt[phase][n] += 1
        if mi >= begins[-1] and e - s >= 30_000:
            big.append((phase, (e - s) / 1e3, n[:160]))
        if mi >= begins[-1] and phase == os.environ.get("PH_SEQ"):
            seqk.append(((e - s) / 1e3, n[:130]))
    K = 3
    print(f"kernel time per phase, mean of the last {K} steps (eager; sum {sum(tot.values()) / 1e6 / K:.2f} ms/step)")
    for ph, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"{v / 1e6 / K:8.3f} ms  {cnt[ph] / K:6.0f} kernels  {ph}")
        for n, kv in kern[ph].most_common(top):
            print(f"            {kv / 1e6 / K:7.3f} {kcnt[ph][n] / K:5.0f}x  {n[:110]}")
    if seqk:
        print(f"\nall kernels of phase {os.environ.get('PH_SEQ')} (last step), in launch order")
        for us, n in seqk:
            print(f"{us:8.1f} us  {n}")
    print("\nkernels >= 30 us of the last step, in launch order")
    for ph, us, n in big:
        print(f"{us:8.1f} us  {ph:22s} {n}")


if __name__ == "__main__":
    {"run": run, "report": report}[sys.argv[1]]()

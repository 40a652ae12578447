#!/usr/bin/env python
"""bench.py -- headline benchmark of the SpaCap3D hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): scenes/sec of one full training step (forward of SpaCapNet + total loss + backward +
single gradient all-reduce + Adam) at cfg2 -- ScanRefer-shaped synthetic scans, 40 000 points (xyz + height),
256 proposals, 8 scenes per GPU, default model (6+6 layers, d_model 128, h 8, d_ff 2048, vocab 3001, relation
head on, attention dropout 0.1 as in the reference) -- fp32, inputs resident in HBM.  One rank per GPU; scenes
are sharded by rank (weak scaling); rank 0 prints ONE JSON line.

Extra objects on the line (tier contract):
  roofline      the kernel FUNCTION with the largest summed duration on the step's own stream, read at run time from the first row of
                the newest window table under profiles/ (rNN_*_timed_window_kernels.csv, tools/prof_window.py over a rocprofv3
                kernel trace of this very command; the line fails loudly when the table is empty or names a function without a
                case here), at that function's LARGEST launch of the step.  launch_us = its duration INSIDE eager training
                steps run right after the timed region (HIP events on the step's stream around the C-ABI call, the next batch's
                sampling chain beside it); achieved = algorithmic flops or bytes / launch_us against the roof that binds the
                implemented arithmetic (tools/kernel_cases.py: roofline_entry).  traffic = HBM bytes per launch from the tracked
                rocprofv3 --pmc passes (profiles/rNN_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, tools/pmc_parse.py).
  roofline_more the same object for: the largest shared-MLP forward layer (SA2 layer 3, 262 144 rows, 128 -> 256, streaming
                split-bf16 kernel, HBM-bound, in-step duration as above), the relation head's forward, the largest
                data-gradient GEMM (SA2 layer 3), an HBM-bound layer (SA1 layer 2, 64 -> 64 on 1 048 576 rows), the encoder's
                feed-forward block, and the SA1 furthest-point sampling, which is neither: an on-chip latency chain (bound
                "latency", us per round).
  step          algorithmic flops of the whole training step and the resulting fraction of the fp32-MFMA peak.
  drop_in       scenes/s when the caller invokes the model unchanged (no Trainer-level pipelining: furthest-point
                sampling, ball queries and interpolation weights computed inside the step).
  cpu_baseline  the same training step on the host CPU (this repo's host code on device "cpu" driving the CPU
                oracle ops, oracle/): single-thread canonical and OpenMP variants, warm-up + >= 3 timed steps each on a
                bounded batch; per-operator CPU rates for FPS / ball query.  rank 0, N = 1 only.
  ops           FPS and ball_query Mpts/s at the SA1 shape (the second half of BASELINE.json's metric).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

# Hardware queues of the HIP runtime (read when the runtime starts: before torch is imported; inherited by the rank / sub-record
# processes).  The captured step, its two forked branches and the next batch's sampling graph run on internal streams that the
# runtime maps onto this many queues; two of them on one queue serialise.  Measured on one box: 3 -> 9.77 ms
# per step, 4 (the runtime's default) -> 6.81, 5 -> 9.78, 6 .. 24 -> 6.79 - 6.83.  Eight is inside the flat region instead of
# between the two cliffs; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start N fresh rank processes --
    one per GPU, the reference's one-command multi-GPU mode (scripts/train.py:198-200) -- BEFORE anything in this process
    has touched the GPU (nothing has been imported yet but the standard library).  The parent never initialises HIP and never
    re-execs; it relays rank 0's JSON line and exits non-zero when any rank fails."""
    import collections
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, tails, out0, threads = [], [], [], []

    def drain(stream, keep, limit=None):
        # a reader thread per pipe: no child can ever block on a full pipe while the parent looks elsewhere
        for line in iter(stream.readline, b""):
            keep.append(line)
            if limit is not None:            # a rank's stderr: passed through as it comes, the last lines kept for the report
                sys.stderr.buffer.write(line)
                sys.stderr.buffer.flush()
                if len(keep) > limit:
                    keep.popleft()
        stream.close()

    # HSA_ENABLE_IPC_MODE_LEGACY=0: this image's host driver only supports dmabuf IPC handles; without it RCCL's intra-node
    # transport set-up fails with `hipIpcGetMemHandle: invalid argument` (the image exports it already: stated here so that a
    # caller with a scrubbed environment gets the same ranks)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE,
                             start_new_session=True)
        procs.append(p)
        tails.append(collections.deque())
        threads.append(threading.Thread(target=drain, args=(p.stderr, tails[-1], 60), daemon=True))
        if r == 0:
            threads.append(threading.Thread(target=drain, args=(p.stdout, out0), daemon=True))
    for t in threads:
        t.start()
    # poll ALL ranks: the first non-zero exit ends the job within seconds (the survivors would otherwise sit in
    # init_process_group / their next collective until the store's 10 - 30 min timeout)
    first_bad = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            first_bad = bad
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    if first_bad:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)      # exactly the process group this function started
                except ProcessLookupError:
                    pass
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.wait()
    for t in threads:
        t.join(timeout=5.0)
    sys.stdout.write(b"".join(out0).decode(errors="replace"))
    sys.stdout.flush()
    if first_bad:
        for r, tail in enumerate(tails):
            text = b"".join(tail).decode(errors="replace")
            sys.stderr.write(f"---- rank {r} (exit code {procs[r].returncode}) last stderr lines ----\n{text}")
    sys.stderr.flush()
    if first_bad:
        raise SystemExit(f"bench.py: rank(s) failed first (rank, exit code): {first_bad}; the other ranks were terminated")


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _ap = argparse.ArgumentParser(add_help=False)
    _ap.add_argument("--gpus", type=int, default=1)
    _n = _ap.parse_known_args()[0].gpus
    if _n > 1:
        spawn_ranks(_n)
        raise SystemExit(0)

import torch  # noqa: E402

from spacap3d_amd import backend, synthetic as S  # noqa: E402
from spacap3d_amd.distributed import init_from_env  # noqa: E402
from spacap3d_amd.engine import Trainer, synthetic_batch  # noqa: E402
from spacap3d_amd.spacapnet import build_default  # noqa: E402

CFG = {  # BASELINE.json configs[1..]
    "cfg2": dict(n_points=40000, proposals=256, batch=8, feats=dict(), transformer=dict()),
    "cfg3": dict(n_points=40000, proposals=256, batch=8, feats=dict(use_color=True, use_normal=True), transformer=dict()),
    "cfg4": dict(n_points=40000, proposals=256, batch=8, feats=dict(use_multiview=True, use_normal=True), transformer=dict()),
    # stress config: 80 000 points, 512 proposals, 16 scenes / GPU; "d_model=512" cannot run in the reference (no input
    # projection, relation head hard-codes d_k = 16: SURVEY.md section 5) -- built here with the documented deviation:
    # Linear(128, 512) token projection (transformer_captioner.TransformerDecoderModel.token_proj) and h = 32
    "cfg5": dict(n_points=80000, proposals=512, batch=16, feats=dict(), transformer=dict(d_model=512, h=32)),
}


class KernelTimer:
    """Brackets one native op with events on the stream it is launched on (our kernels are enqueued on
    torch's current stream, so torch.cuda.Event sees them)."""

    def __init__(self, fn, select):
        self.fn, self.select, self.events, self.on = fn, select, [], False

    def __call__(self, *a, **k):
        if self.on and self.select(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.fn(*a, **k)
            e1.record()
            self.events.append((e0, e1))
            return out
        return self.fn(*a, **k)

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))


def time_op(fn, iters=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _cpu_steps(cfg, be, threads, batch, steps):
    """warm-up step at size + `steps` timed steps of the training step on the CPU with backend `be`."""
    torch.set_num_threads(threads)
    with backend.use_backend(be):
        torch.manual_seed(0)
        model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"],
                              **cfg["transformer"])
        model.train()
        tr = Trainer(model, S.mean_size_arr().numpy())
        data = synthetic_batch(batch, cfg["n_points"], "cpu", seed=0, **cfg["feats"])
        tr.step(data)   # warm-up AT SIZE (optimizer set-up, thread pools, allocator), untimed
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            tr.step(data)
            ts.append(time.perf_counter() - t0)
    return ts


def _cpu_ops(ext, B, N, m=2048, radius=0.2, nsample=64):
    xyz = S.scene_batch(B, N, use_height=False, seed=1000)
    t0 = time.perf_counter()
    inds = ext.furthest_point_sampling(xyz, m)
    t_fps = time.perf_counter() - t0
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    t0 = time.perf_counter()
    ext.ball_query(new_xyz, xyz, radius, nsample)
    t_bq = time.perf_counter() - t0
    return {"scenes": B, "fps_Mpts_s": B * N / t_fps * 1e-6, "fps_s": t_fps, "ball_query_Mpts_s": B * N / t_bq * 1e-6,
            "ball_query_s": t_bq}


def cpu_baseline(cfg, omp_batch, steps, threads=8):
    """the same training step on the host CPU, this repo's host code on device "cpu" with the C oracle
    ops + torch CPU (kind "port"), (i) single thread, canonical oracle; (ii) OpenMP oracle + all torch threads."""
    from oracle.attention_ref import OracleBackend
    omp = OracleBackend(openmp=True)
    # torch's CPU kernels stop scaling early on this model (measured on the MI355X box's 128-thread host, B = 2:
    # 8 threads 1.52 scenes/s, 16: 1.44, 32: 1.24, 64: 0.68, 128: 0.34, 1: 0.74; tools/lab/cpu_threads.py): the multi-core
    # variant uses the best count, not every hardware thread
    host_threads = os.cpu_count() or 1
    cores = min(host_threads, threads)
    t_omp = _cpu_steps(cfg, omp, cores, omp_batch, steps)
    # (torch.set_num_threads also sets the OpenMP thread count the oracle's OpenMP build runs with)
    torch.set_num_threads(min(host_threads, 8))
    ops_omp = _cpu_ops(omp._ext, 8, cfg["n_points"])          # parallel over the 8 scenes / the centres
    ops_omp["threads"] = min(host_threads, 8)
    t_one = _cpu_steps(cfg, OracleBackend(openmp=False), 1, 1, steps)
    ops_one = _cpu_ops(OracleBackend(openmp=False)._ext, 2, cfg["n_points"])
    torch.set_num_threads(host_threads)
    mean = lambda v: sum(v) / len(v)
    return {"value": omp_batch / mean(t_omp), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"OpenMP variant: 1 warm-up + {steps} timed full training steps on B = {omp_batch} synthetic scenes of "
                      f"{cfg['n_points']} points ({', '.join(f'{t:.1f}' for t in t_omp)} s): this repo's host code on device cpu + "
                      f"the C oracle ops (OpenMP over scene / centre) + torch CPU with {cores} threads (the fastest count on this "
                      f"host of {host_threads} hardware threads)",
            "single_thread": {"value": 1 / mean(t_one), "unit": "scenes/s", "cores": 1,
                              "sample": f"canonical single-thread oracle + torch with 1 thread: 1 warm-up + {steps} timed steps on "
                                        f"B = 1 ({', '.join(f'{t:.1f}' for t in t_one)} s)"},
            "ops": {"single_thread": ops_one, "openmp": ops_omp,
                    "what": "oracle FPS N -> 2048 and ball query 2048 x N (r 0.2, 64 samples), one call each"}}



def quick_config(name, rank, dev, steps=5, warmup=5):
    """A short pipelined run of another BASELINE configuration (driver-visible sub-record of the bench line): same step as
    the headline (prefetch on the side stream, hipGraph replay), `steps` timed steps after `warmup`."""
    cfg = CFG[name]
    torch.manual_seed(0)
    model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"],
                          **cfg["transformer"]).to(dev)
    model.train()
    tr = Trainer(model, S.mean_size_arr().numpy())
    data = synthetic_batch(cfg["batch"], cfg["n_points"], dev, seed=1000 + rank, **cfg["feats"])
    tr.step(data, next_data=data)
    graphed = tr.enable_graph(data)
    for _ in range(warmup):
        tr.step(data, next_data=data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(data, next_data=data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rec = {"value": cfg["batch"] / dt, "unit": "scenes/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": warmup,
           "scenes_per_gpu": cfg["batch"], "points": cfg["n_points"], "proposals": cfg["proposals"],
           "extra_channels": S.num_extra_channels(**cfg["feats"]), "hip_graph": bool(graphed), "final_loss": float(loss),
           "n_gpus": 1, "note": "per-GPU shard of the configuration on ONE GPU (BASELINE.json quotes it on 4 - 8 GPUs)"}
    if cfg["transformer"]:
        rec["transformer"] = dict(cfg["transformer"], deviation="Linear(128, d_model) token projection, SURVEY.md section 5")
    del tr, model, data
    import gc
    gc.collect()               # (Trainer / autograd reference cycles keep captured graphs and their memory pools alive: the next
    torch.cuda.empty_cache()   # configuration then allocates out of fragmented leftovers -- cfg4 ran 12.7 instead of 9.0 ms)
    return rec


def quick_config_in_child(name, timeout=900):
    """`quick_config(name)` in a FRESH process (this file with --quick-config): the record of a configuration as a user training it
    gets it -- one Trainer per process.  Inside the process that has already trained cfg2 the same step replays 2 - 3 ms slower for
    ONE of the following configurations (which one depends on what ran before: cfg4 after the in-step timers, cfg3 without them;
    tools/lab/quick_config_steps.py): the runtime places the internal streams of a new graph on the hardware queues that earlier
    graphs of the process left least used, and the step's forked branches then share a queue with the sampling chain
    (DESIGN.md section 5).  The parent is idle while the child runs."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--quick-config", name]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout,
                             env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    except subprocess.TimeoutExpired:
        raise SystemExit(f"bench.py: sub-record {name} did not finish within {timeout} s")
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        raise SystemExit(f"bench.py: sub-record {name} failed (exit code {out.returncode}):\n" + "\n".join(out.stderr.splitlines()[-15:]))
    rec = json.loads(lines[-1])
    rec["process"] = "fresh child process (one Trainer per process)"
    return rec


def eval_record(model, data, iters=5):
    """Inference forward: detector + encoder once + greedy decoding of B*K captions for 31 steps
    (models/transformer_captioner.py:402-453), no gradients; model.eval().  Two figures: a stream of batches through
    engine.Evaluator (the next batch's sampling pyramid on a side stream while this one decodes, as the training step is
    benchmarked) and the single, unpipelined forward."""
    from spacap3d_amd.engine import Evaluator
    was = model.training
    model.eval()
    d = {k: v for k, v in data.items() if k != "_fps_prefetch"}
    with torch.no_grad():
        out = model(dict(d), is_eval=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(dict(d), is_eval=True)
        torch.cuda.synchronize()
        dt_single = (time.perf_counter() - t0) / iters
        ev = Evaluator(model)
        cur = dict(d)
        for _ in range(2):
            out = ev(cur, next_data=cur)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = ev(cur, next_data=cur)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    model.train(was)
    B, K = out["lang_cap"].shape[:2]
    return {"ms_per_forward": dt * 1e3, "ms_per_forward_unpipelined": dt_single * 1e3, "scenes_per_s": B / dt, "captions_per_s": B * K / dt,
            "scenes": B, "captions": B * K, "decode_steps": int(out["lang_cap"].shape[2]), "iters": iters,
            "what": "SpaCapNet eval forward: detector + 6-layer encoder once + greedy decoding of B*K captions (key / value cache, "
                    "fused decode step, word choice without logits), host-timed incl. launches.  ms_per_forward: batches in a "
                    "stream, the next batch's sampling / grouping pyramid computed on a side stream while this one decodes "
                    "(engine.Evaluator; every forward still computes one pyramid); ms_per_forward_unpipelined: one forward with its "
                    "own 4.5 ms sampling chain on the critical path"}


class InStepTimer:
    """Brackets the calls of ONE C entry point that match `select(args)` with HIP events on the stream they are launched on
    (the step's own stream): the duration the roofline kernel has INSIDE a training step, beside the side-stream sampling
    chain -- measured over a few EAGER steps after the timed region (inside the replayed hipGraph events cannot bracket a
    kernel; the kernels and what runs beside them are the same)."""

    def __init__(self, mod, name, select):
        self.mod, self.name, self.select, self.events = mod, name, select, []
        self.fn = getattr(mod.lib, name)

    def __enter__(self):
        timer = self

        class _Lib:
            def __getattr__(self_, k):
                f = getattr(timer.mod_lib, k)
                if k != timer.name:
                    return f

                def wrapped(*a):
                    if not timer.select(a):
                        return f(*a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = f(*a)
                    e1.record()
                    timer.events.append((e0, e1))
                    return rc
                return wrapped
        self.mod_lib = self.mod.lib
        self.mod.lib = _Lib()
        return self

    def __exit__(self, *exc):
        self.mod.lib = self.mod_lib
        return False

    def mean_us(self):
        return sum(a.elapsed_time(b) for a, b in self.events) * 1e3 / max(1, len(self.events))


def window_table_top():
    """(function name, table file) of the first row of the newest profiles/rNN_*_timed_window_kernels.csv: the kernel function
    with the largest summed main-stream duration in the timed steps (tools/prof_window.py).  Raises when there is no usable
    table: the roofline object must name what the profile says, not what a comment remembers."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_timed_window_kernels.csv")))
    # the headline workload's tables only (rNN_cfg3_... / rNN_eval_... are the other configurations' evidence)
    files = [f for f in files if os.path.getsize(f) > 0 and "_cfg" not in os.path.basename(f) and "_eval" not in os.path.basename(f)]
    if not files:
        raise SystemExit("bench.py: no non-empty profiles/r*_timed_window_kernels.csv (run tools/prof_step.sh and commit its table)")
    path = files[-1]
    rows, on = [], False
    for line in open(path):
        if line.startswith("main_ms/step,calls/step,avg_us,function"):
            on = True
            continue
        if on:
            if line.startswith("#") or not line.strip():
                break
            rows.append(next(csv.reader([line])))
    if not rows:
        raise SystemExit(f"bench.py: {path} holds no per-function rows")
    return rows[0][3], os.path.basename(path), [(r[3], float(r[0]), float(r[1])) for r in rows[:8]]


def step_flops(cfg, B):
    """Algorithmic flops of one training step (SURVEY.md section 8d formulas): dense layers 2*cin*cout*rows forward, x3 for
    forward + data gradient + weight gradient (first layers of SA1 / the embedding have no data gradient: < 1 %)."""
    N, P = cfg["n_points"], cfg["proposals"]
    C = S.num_extra_channels(**cfg["feats"])
    d = cfg["transformer"].get("d_model", 128)
    dff, V, Lw = 2048, 3001, 32
    mlp = lambda rows, chans: sum(2.0 * a * b * rows for a, b in zip(chans[:-1], chans[1:]))
    f = mlp(2048 * 64, [C + 3, 64, 64, 128]) + mlp(1024 * 32, [131, 128, 128, 256]) + mlp(512 * 16, [259, 128, 128, 256])
    f += mlp(256 * 16, [259, 128, 128, 256]) + mlp(512, [512, 256, 256]) + mlp(1024, [512, 256, 256])
    f += mlp(1024, [256, 256, 256, 259]) + mlp(P * 16, [259, 128, 128, 128]) + mlp(P, [128, 128, 128, 97])
    enc_layer = lambda L: 2.0 * L * d * d * 4 + 4.0 * L * L * d + 2.0 * L * d * dff * 2
    f += 6 * enc_layer(P) + 6 * enc_layer(Lw) + 2.0 * (Lw - 1) * d * V + (2.0 * P * 128 * d if d != 128 else 0.0)
    f += 2.0 * P * P * (d * d * 2 + d * 9)                      # relation head
    return 3.0 * f * B

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg2", choices=sorted(CFG))
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default: the config's 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="sample inside the step instead of one step ahead")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--ablate", default="", help="analysis only (NOT the headline metric): 'relation' drops the "
                                                 "relation head, 'caption' the whole captioner")
    ap.add_argument("--cpu-sample", type=int, default=4, help="scenes per step in the multi-core CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=8, help="torch threads of the multi-core CPU baseline (8 is the fastest on the box's host)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (after one warm-up at size)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the extra unpipelined (drop-in caller) measurement")
    ap.add_argument("--no-in-step", action="store_true", help="skip the eager steps that time the roofline kernel inside a step "
                                                              "(for rocprofv3 runs: tools/prof_window.py takes the LAST steps of the trace)")
    ap.add_argument("--quick-config", default=None, choices=sorted(CFG), help="internal: print quick_config(NAME) as one JSON line "
                    "and exit (how the default line's cfg3 / cfg4 / cfg5 sub-records are produced, each in its own process)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short cfg3 / cfg4 / cfg5 sub-records and the eval record")
    args = ap.parse_args()
    if args.quick_config:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(0)
        print(json.dumps(quick_config(args.quick_config, 0, torch.device("cuda", 0))), flush=True)
        return

    # (read and validated on EVERY rank before any rank joins the process group: a bad table ends all ranks at once instead
    # of leaving the others in their next collective)
    top_fn, top_file, top_rows = window_table_top()
    if os.environ.get("SPACAP_BENCH_FAIL_RANK") is not None and os.environ.get("SPACAP_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit("bench.py: injected failure of this rank (test knob SPACAP_BENCH_FAIL_RANK)")
    rank, local_rank, world = init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("SPACAP_SHARE_GPU") == "1":   # test knob: all ranks on the GPUs that exist (with SPACAP_DIST_BACKEND=gloo)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = CFG[args.config]
    per_gpu = args.batch or cfg["batch"]

    be = backend.ops()  # HIP backend; raises if the extension is not built
    fps_timer = KernelTimer(be.furthest_point_sampling, lambda pts, m: pts.shape[1] == cfg["n_points"])
    be.furthest_point_sampling = fps_timer

    torch.manual_seed(0)
    model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"],
                          **cfg["transformer"]).to(dev)
    model.train()
    if args.ablate == "relation":
        model.caption.check_relation = False
        model.caption.model.encoder.layers[-1].self_attn.keep_value = False
    trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=(args.ablate != "relation"))
    # each rank owns its own shard of scenes (seed + rank), resident in HBM before the timed region
    data = synthetic_batch(per_gpu, cfg["n_points"], dev, seed=1000 + rank, **cfg["feats"])

    # Every step starts the furthest-point-sampling pyramid of the NEXT batch on a side stream (here the next
    # batch is the same resident tensor, but it is recomputed every step: K timed steps = K pyramids).
    nxt = None if args.no_prefetch else data
    trainer.step(data, next_data=nxt)  # eager: MIOpen solver search, optimizer / bucket set-up, first prefetch
    graphed = False
    if not args.no_graph:
        graphed = trainer.enable_graph(data)
        if not graphed and rank == 0:
            print(f"[bench] hipGraph capture failed, running eagerly: {trainer.graph_error}", file=sys.stderr)
        if not graphed and nxt is not None and "_fps_prefetch" not in data:
            trainer.prefetch(data)
    for _ in range(max(1, args.warmup)):
        trainer.step(data, next_data=nxt)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    fps_timer.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(data, next_data=nxt)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fps_timer.on = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)
    n_params, allreduce_bytes = sum(p.numel() for p in model.parameters()), trainer.bucket.nbytes

    # -- the roofline kernel's duration INSIDE a step (beside the side-stream sampling chain, with the step's grid) ----------
    in_step = in_step_top = None
    # (function of the window table) -> (module whose `lib` launches it, C entry point, selector of its LARGEST launch in the step)
    R1_, R2_ = per_gpu * 2048 * 64, per_gpu * 1024 * 32
    import spacap3d_amd.linear as lin
    import spacap3d_amd.sa_mlp as sam
    import spacap3d_amd.tf_layer as tfl
    TOP = {
        # (SA1 layer 2: the pooled layer 3 runs sa_wgrad_pool_kernel, so (R1, 64, 64) is the largest launch of THIS function in the step)
        "sa_wgrad_kernel": (sam, "spacap_sa_wgrad_f32", lambda a: (a[7], a[8], a[9]) == (R1_, 64, 64)),
        "rel_fused_bwd_kernel": (lin, "spacap_relation_fused_bwd_f32", lambda a: True),
        "sa_mid_fwd_bf3s_kernel": (sam, "spacap_sa_mid_fwd_pool_f32", lambda a: (a[4], a[5], a[6]) == (R2_, 128, 256)),
        "sa_dgrad_bf3s_kernel": (sam, "spacap_sa_dgrad_f32", lambda a: (a[8], a[9], a[10]) == (R2_, 256, 128)),
        "tf_ffn_kernel": (tfl, "spacap_tf_ffn_f32", lambda a: a[0] == 0),
        "tf_ffn_bf3_kernel": (tfl, "spacap_tf_ffn_bf3_f32", lambda a: a[0] == 0 and a[5] == per_gpu * 256),
        "tf_rows_kernel": (tfl, "spacap_tf_rows_f32", lambda a: (a[0]._obj.mode, a[0]._obj.R, a[0]._obj.n2) == (0, per_gpu * 256, 384)
                           and a[0]._obj.nparts > 0),
    }
    if world == 1 and nxt is not None and not args.ablate and not args.no_in_step and (per_gpu, cfg["n_points"]) == (8, 40000):
        if top_fn not in TOP:
            raise SystemExit(f"bench.py: the window table {top_file} names `{top_fn}` as the largest main-stream kernel function, "
                             f"for which bench.py has no roofline case: add one (TOP / tools/kernel_cases.py)")
        keep_graph, trainer.graph = trainer.graph, None     # a few EAGER steps: same kernels, events can bracket them
        tmod, tname, tsel = TOP[top_fn]
        with InStepTimer(sam, "spacap_sa_mid_fwd_pool_f32", TOP["sa_mid_fwd_bf3s_kernel"][2]) as ist, \
                InStepTimer(tmod, tname, tsel) as itop:
            for _ in range(8):
                trainer.step(data, next_data=nxt)
            torch.cuda.synchronize()
        if ist.events:
            in_step = {"us": ist.mean_us(), "launches": len(ist.events)}
        if not itop.events:
            # the roofline line must price a launch the step really makes, never an isolated stand-in for one it does not
            raise SystemExit(f"bench.py: eight eager steps launched `{top_fn}` ({tname}) at no shape its TOP selector accepts: "
                             "the selector / case no longer match what the step runs")
        in_step_top = {"us": itop.mean_us(), "launches": len(itop.events)}
        trainer.graph = keep_graph

    # -- inference forward (greedy decoding) -------------------------------------------------------------------------------
    eval_rec = None
    if world == 1 and rank == 0 and not args.no_configs and not args.ablate:
        eval_rec = eval_record(model, data)

    # -- drop-in caller: the model invoked unchanged, everything (sampling, grouping, neighbour search) inside the step --
    drop_in = None
    if world == 1 and nxt is not None and not args.no_drop_in and not args.ablate:
        data.pop("_fps_prefetch", None)
        ok = True
        if graphed:
            trainer.graph = None
            ok = trainer.enable_graph(data, warmup=1)   # captured WITHOUT a pyramid: the graph samples in line
        for _ in range(5):
            trainer.step(data)
        torch.cuda.synchronize()
        k2 = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(k2):
            trainer.step(data)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / k2
        drop_in = {"value": per_gpu / dt2, "unit": "scenes/s", "ms_per_step": dt2 * 1e3, "steps": k2, "hip_graph": bool(ok and graphed),
                   "what": "same training step without Trainer-level pipelining: furthest-point sampling, ball queries and "
                           "interpolation weights computed inside the step (what a caller of SpaCapNet.forward gets)"}

    trainer_prefetch_graph = bool(getattr(trainer, "prefetch_graph", False))
    reserved_cus = int(per_gpu) if nxt is not None else 0
    if rank == 0:
        import glob
        import kernel_cases as KC
        ms_per_step = dt / args.steps * 1e3
        B, N, m = per_gpu, cfg["n_points"], 2048
        del trainer
        torch.cuda.empty_cache()
        # rocprofv3 --pmc evidence collected by tools/pmc_run.sh at the cfg2 shapes and committed under profiles/
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
        pmc = json.load(open(pmc_files[-1])) if (pmc_files and (B, N) == (8, 40000)) else {}
        R2, R1 = B * 1024 * 32, B * 2048 * 64
        # the layer kernels are timed with their full grid (nothing runs beside them here); the step launches the forward
        # ones with `reserved_cus` CUs left to the sampling chain: that launch is timed as well
        KC.check(KC.lib.spacap_sa_reserve_cus(0), "spacap_sa_reserve_cus")
        # `roofline` = the kernel function with the largest summed duration on the step's own stream = the first row of the
        # window table under profiles/ (read above), at its largest launch of the step, priced with the duration it has INSIDE
        # the step and against the roof that binds the implemented arithmetic
        how_tail = ("launch_us = mean duration of this launch INSIDE {n} eager training steps run right after the timed region (HIP "
                    "events on the step's stream around the C-ABI call; the sampling chain of the next batch runs beside it on the "
                    "side stream, as in the replayed step -- compare the kernel's row in profiles/*_step_timeline.txt); "
                    "launch_us_isolated_*: 20 back-to-back launches with nothing beside them; traffic from "
                    + (os.path.basename(pmc_files[-1]) if pmc else "no tracked PMC file for this shape"))
        R1 = B * 2048 * 64
        top_case = {
            "sa_wgrad_kernel": lambda: KC.sa_wgrad(R1, 64, 64, False, 64, dev, "SA1 layer 2"),
            "rel_fused_bwd_kernel": lambda: KC.rel_fused(B, cfg["proposals"], 1, dev),
            "sa_mid_fwd_bf3s_kernel": lambda: KC.sa_mid_fwd(R2, 128, 256, dev, "SA2 layer 3"),
            "sa_dgrad_bf3s_kernel": lambda: KC.sa_dgrad(R2, 256, 128, True, 32, dev, "SA2 layer 3"),
            "tf_ffn_kernel": lambda: KC.tf_ffn(256, 2048, 0, dev),
            "tf_ffn_bf3_kernel": lambda: KC.tf_ffn(B * 256, 2048, 0, dev),
            "tf_rows_kernel": lambda: KC.tf_rows(B * cfg["proposals"], cfg["transformer"].get("d_ff", 2048), dev),
        }
        if top_fn not in top_case:
            raise SystemExit(f"bench.py: no roofline case for `{top_fn}` (first row of {top_file})")
        c_top = top_case[top_fn]()
        iso_top = KC.time_case(c_top)
        roof = KC.roofline_entry(c_top, in_step_top["us"] if in_step_top else iso_top, pmc)
        roof["launch_us_isolated_full_grid"] = iso_top
        roof["function"] = top_fn
        roof["chosen_by"] = (f"first row of profiles/{top_file} (largest summed main-stream duration per step: "
                             + "; ".join(f"{n} {ms:.3f} ms in {c:.0f} launches" for n, ms, c in top_rows[:4]) + ")")
        if in_step_top:
            roof["in_step_launches_timed"] = in_step_top["launches"]
        roof["how"] = how_tail.format(n=in_step_top["launches"] if in_step_top else 0)
        del c_top
        c_rel = KC.rel_fused(B, cfg["proposals"], 1, dev)
        roof_relb = KC.roofline_entry(c_rel, KC.time_case(c_rel), pmc)
        del c_rel
        c_relf = KC.rel_fused(B, cfg["proposals"], 0, dev)
        roof_relf = KC.roofline_entry(c_relf, KC.time_case(c_relf), pmc)
        del c_relf
        # the shared-MLP layer kernel with the largest summed duration (sa_mid_fwd_bf3s_kernel<128, .., pooled>, four launches
        # per step) at its largest launch (SA2 layer 3): HBM-bound as split-bf16
        c_mfma = KC.sa_mid_fwd(R2, 128, 256, dev, "SA2 layer 3")
        iso_us = KC.time_case(c_mfma)
        roof_sa = KC.roofline_entry(c_mfma, in_step["us"] if in_step else iso_us, pmc)
        roof_sa["launch_us_isolated_full_grid"] = iso_us
        if in_step:
            roof_sa["in_step_launches_timed"] = in_step["launches"]
        if reserved_cus:
            KC.check(KC.lib.spacap_sa_reserve_cus(reserved_cus), "spacap_sa_reserve_cus")
            roof_sa["launch_us_isolated_with_the_steps_grid"] = KC.time_case(c_mfma)
            KC.check(KC.lib.spacap_sa_reserve_cus(0), "spacap_sa_reserve_cus")
        roof_sa["how"] = how_tail.format(n=in_step["launches"] if in_step else 0)
        del c_mfma
        c_hbm = KC.sa_mid_fwd_l1in(R1, dev, "SA1 layer 2")
        roof_hbm = KC.roofline_entry(c_hbm, KC.time_case(c_hbm), pmc)
        del c_hbm
        c_dg = KC.sa_dgrad(R2, 256, 128, True, 32, dev, "SA2 layer 3")
        roof_dg = KC.roofline_entry(c_dg, KC.time_case(c_dg), pmc)
        del c_dg
        c_ffn = KC.tf_ffn(B * 256, 2048, 0, dev)
        roof_ffn = KC.roofline_entry(c_ffn, KC.time_case(c_ffn), pmc)
        c_wp = KC.sa_wgrad_pool(R1, 64, 128, 64, dev, "SA1 layer 3")
        roof_wp = KC.roofline_entry(c_wp, KC.time_case(c_wp), pmc)
        del c_wp
        del c_ffn
        c_fps = KC.fps(B, N, m, dev)
        t_fps_us = KC.time_case(c_fps, iters=5, warm=1)
        roof_fps = KC.roofline_entry(c_fps, t_fps_us, pmc)
        roof_fps["in_step_us"] = fps_timer.mean_ms() * 1e3 if fps_timer.events else None
        roof_fps["in_step_launches_timed"] = len(fps_timer.events)
        if not fps_timer.events:
            roof_fps["in_step_note"] = ("the next batch's pyramid replays as ONE side-stream graph beside the step (Trainer.prefetch), so "
                                        "events cannot bracket the sampling kernel; its in-step duration is in the rocprofv3 window "
                                        "table under profiles/ (fps_bucket_kernel); SPACAP_PREFETCH_GRAPH=0 launches it eagerly")
        roof_fps["compulsory_bytes"] = c_fps["compulsory_bytes"]
        del c_fps
        # isolated SA1-shaped op rates (second half of the BASELINE metric)
        xyz = data["point_clouds"][..., :3].contiguous()
        t_fps = time_op(lambda: fps_timer.fn(xyz, m))
        inds = fps_timer.fn(xyz, m)
        new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        t_bq = time_op(lambda: be.ball_query(new_xyz, xyz, 0.2, 64))
        ops = {"fps_Mpts_s": B * N / t_fps * 1e-3, "fps_ms": t_fps, "fps_Gupdates_s": B * N * (m - 1) / t_fps * 1e-6,
               "ball_query_Mpts_s": B * N / t_bq * 1e-3, "ball_query_ms": t_bq,
               "ball_query_Gpairs_s": B * m * N / t_bq * 1e-6}
        fl = step_flops(cfg, per_gpu)
        line = {
            "metric": "scenes/sec (40k pts, 256 proposals) fwd+bwd", "value": per_gpu * world / (dt / args.steps),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic ScanRefer-shaped scans, {N} pts x (xyz+"
                                   f"{S.num_extra_channels(**cfg['feats'])} ch), {cfg['proposals']} proposals, "
                                   f"{per_gpu} scenes/GPU; full training step (SpaCapNet fwd + loss + bwd + "
                                   f"grad all-reduce + Adam)"
                                   + (f"; transformer {cfg['transformer']} with the 128->512 token projection" if cfg["transformer"] else ""),
                       "global_batch": per_gpu * world, "parallelism": f"dp{world}",
                       "hip_graph": bool(graphed), "fps_prefetch_side_stream": nxt is not None,
                       "prefetch_as_graph": bool(graphed and nxt is not None and trainer_prefetch_graph), "reserved_cus_forward": reserved_cus, "geometry_prefetch": nxt is not None, "deferred_weight_gradients": True,
                       "sa_forward_gemm": ("fp32 MFMA (v_mfma_f32_16x16x4_f32)" if os.environ.get("SPACAP_SA_F32MFMA", "0") not in ("", "0") else
                                           "split-bf16 x3 streaming kernel: 6 bf16 MFMA products per fp32 product, fp32 accumulate"),
                       "params": n_params, "allreduce_bytes": allreduce_bytes},
            "roofline": roof, "roofline_more": [roof_sa, roof_relb, roof_relf, roof_dg, roof_hbm, roof_ffn, roof_wp, roof_fps],
            "step": {"algorithmic_flops": fl, "achieved_TFLOPs": fl / (ms_per_step * 1e-3) * 1e-12,
                     "frac_of_fp32_mfma_peak": fl / (ms_per_step * 1e-3) * 1e-12 / KC.PEAK_MFMA_F32_TFLOPS},
            "ops": ops, "final_loss": loss_val,
        }
        if drop_in is not None:
            line["drop_in"] = drop_in
        if eval_rec is not None:
            line["eval"] = eval_rec
        if world == 1 and args.config == "cfg2" and not args.no_configs and not args.ablate:
            del model, data
            trainer = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line["configs"] = {name: quick_config_in_child(name) for name in ("cfg3", "cfg4", "cfg5")}
        if args.ablate:
            line["metric"] += f" [ABLATION {args.ablate}: not the headline metric]"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample, args.cpu_steps, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- headline benchmark of the SpaCap3D hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): scenes/sec of one full training step (forward of SpaCapNet + total loss + backward +
single gradient all-reduce + Adam) at cfg2 -- ScanRefer-shaped synthetic scans, 40 000 points (xyz + height),
256 proposals, 8 scenes per GPU, default model (6+6 layers, d_model 128, h 8, d_ff 2048, vocab 3001, relation
head on, attention dropout 0.1 as in the reference) -- fp32, inputs resident in HBM.  One rank per GPU; scenes
are sharded by rank (weak scaling); rank 0 prints ONE JSON line.

Extra objects on the line (tier contract):
  roofline      the dominant hand-written kernel (SA1 furthest-point sampling, N -> 2048): algorithmic bytes
                B*(m-1)*N*20 (SURVEY.md section 8d streamed-traffic model) / its average duration measured with
                events on the stream it is launched on (the prefetch side stream, i.e. while it shares the chip with
                the dense kernels of the step) inside the timed steps, against 8 TB/s HBM; `isolated_ms` is the same
                launch alone on an idle chip.
  cpu_baseline  the same training step on the host CPU (this repo's host code on device "cpu" driving the CPU
                oracle ops, oracle/) on a bounded sample; rank 0, N = 1 only.
  ops           FPS and ball_query Mpts/s at the SA1 shape (the second half of BASELINE.json's metric).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


def cpu_baseline(cfg, sample_batch):
    """One training step of the same model on the host CPU with the oracle ops (OpenMP build)."""
    from oracle.attention_ref import OracleBackend
    be = OracleBackend(openmp=True)
    cores = be._ext.num_threads()
    torch.set_num_threads(cores)
    with backend.use_backend(be):
        torch.manual_seed(0)
        model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"],
                              **cfg["transformer"])
        model.train()
        tr = Trainer(model, S.mean_size_arr().numpy())
        small = synthetic_batch(1, 2048, "cpu", seed=1, **cfg["feats"])
        tr.step(small)  # builds the optimizer / thread pools on a tiny input (untimed)
        data = synthetic_batch(sample_batch, cfg["n_points"], "cpu", seed=0, **cfg["feats"])
        t0 = time.perf_counter()
        tr.step(data)
        dt = time.perf_counter() - t0
    return {"value": sample_batch / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"1 full training step on {sample_batch} synthetic scene(s) of {cfg['n_points']} points "
                      f"({dt:.1f} s): this repo's host code on device cpu + the C oracle ops (OpenMP) + torch CPU"}


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
    ap.add_argument("--streams", action="store_true", help="run the detection losses as a side-stream branch (slower since they are fused)")
    ap.add_argument("--ablate", default="", help="analysis only (NOT the headline metric): 'relation' drops the "
                                                 "relation head, 'caption' the whole captioner")
    ap.add_argument("--cpu-sample", type=int, default=2, help="scenes in the CPU-baseline sample")
    args = ap.parse_args()

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
    trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=(args.ablate != "relation"),
                      multi_stream=args.streams)
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

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        B, N, m = per_gpu, cfg["n_points"], 2048
        fps_bytes = B * (m - 1) * N * 20  # 12 B xyz + 4 B temp read + 4 B temp write per point-update
        fps_ms = fps_timer.mean_ms()
        if not fps_ms:   # --no-prefetch with a hipGraph: the launch sits inside the graph, no live events; time it alone
            xyz0 = data["point_clouds"][..., :3].contiguous()
            fps_ms = time_op(lambda: fps_timer.fn(xyz0, m))
        roof = {"bound": "hbm", "kernel": "fps_bucket_kernel<10> (SA1 furthest point sampling, 40000 -> 2048, runs on the prefetch side stream)",
                "achieved": fps_bytes / (fps_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": fps_bytes / (fps_ms * 1e-3) / 1e9 / 8000.0,
                # HBM bytes per launch from rocprofv3 PMC passes (profiles/r01_pmc_fps_*.csv: FETCH_SIZE 4037 KB,
                # doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE 3904 KB); cannot be collected
                # from inside this process.  It is 0.1 % of the algorithmic bytes: the kernel is on-chip resident.
                "traffic": (2 * 4037 + 3904) * 1024 if (B, N) == (8, 40000) else None,
                "launch_ms": fps_ms, "launches_timed": len(fps_timer.events),
                "algorithmic_bytes_per_launch": fps_bytes}
        roof_fill = roof
        # isolated SA1-shaped op rates (second half of the BASELINE metric)
        xyz = data["point_clouds"][..., :3].contiguous()
        t_fps = time_op(lambda: fps_timer.fn(xyz, m))
        inds = fps_timer.fn(xyz, m)
        new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        t_bq = time_op(lambda: be.ball_query(new_xyz, xyz, 0.2, 64))
        roof_fill["isolated_ms"] = t_fps
        roof_fill["isolated_frac"] = fps_bytes / (t_fps * 1e-3) / 1e9 / 8000.0
        ops = {"fps_Mpts_s": B * N / t_fps * 1e-3, "fps_ms": t_fps, "fps_Gupdates_s": B * N * (m - 1) / t_fps * 1e-6,
               "ball_query_Mpts_s": B * N / t_bq * 1e-3, "ball_query_ms": t_bq,
               "ball_query_Gpairs_s": B * m * N / t_bq * 1e-6}
        line = {
            "metric": "scenes/sec (40k pts, 256 proposals) fwd+bwd", "value": per_gpu * world / (dt / args.steps),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic ScanRefer-shaped scans, {N} pts x (xyz+"
                                   f"{S.num_extra_channels(**cfg['feats'])} ch), {cfg['proposals']} proposals, "
                                   f"{per_gpu} scenes/GPU; full training step (SpaCapNet fwd + loss + bwd + "
                                   f"grad all-reduce + Adam)",
                       "global_batch": per_gpu * world, "parallelism": f"dp{world}",
                       "hip_graph": bool(graphed), "fps_prefetch_side_stream": nxt is not None, "geometry_prefetch": nxt is not None, "deferred_weight_gradients": True,
                       "side_stream_branches": bool(trainer.multi_stream),
                       "params": sum(p.numel() for p in model.parameters()),
                       "allreduce_bytes": trainer.bucket.nbytes},
            "roofline": roof, "ops": ops, "final_loss": loss_val,
        }
        if args.ablate:
            line["metric"] += f" [ABLATION {args.ablate}: not the headline metric]"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

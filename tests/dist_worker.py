"""Worker of the multi-rank tests (launched by torch.distributed.run, one process per rank): the REAL training step
(spacap3d_amd.engine.Trainer) on each rank's own scenes, then checks that hold on every rank:
  * the reduced gradient bucket equals the mean of the ranks' local buckets (all-gathered) and is bit-identical on all ranks;
  * after the optimizer steps every rank holds bit-identical parameters (they started from rank 0's broadcast).
--device cpu: gloo + the CPU checker backend (runs anywhere).  --device cuda: RCCL, one GPU per rank (needs >= 2 GPUs).
Prints one line `DIST_OK rank=<r> ...` per rank on success."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="all-reduce the captioner's slice beside the detector's backward")
    ap.add_argument("--share-gpu", action="store_true", help="--device cuda with every rank on cuda:0 over gloo (1-GPU boxes)")
    a = ap.parse_args()
    from spacap3d_amd import backend, synthetic as S
    from spacap3d_amd.distributed import init_from_env
    from spacap3d_amd.engine import Trainer, synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    rank, local_rank, world = init_from_env("gloo" if (a.device == "cpu" or a.share_gpu) else "nccl")
    if a.share_gpu:
        local_rank = 0
    assert world >= 2
    if a.device == "cpu":
        from oracle.attention_ref import OracleBackend
        be, dev = OracleBackend(), torch.device("cpu")
        torch.set_num_threads(2)
    else:
        torch.cuda.set_device(local_rank)
        be, dev = backend.HipBackend(), torch.device("cuda", local_rank)
    with backend.use_backend(be):
        torch.manual_seed(100 + rank)                       # ranks start DIFFERENT; the Trainer's broadcast aligns them
        model = build_default(vocab_size=60, num_proposal=32, N=1, d_ff=128).to(dev).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-3)
        tr.overlap_allreduce = a.overlap
        data = synthetic_batch(1, 2048, dev, seed=7 + rank, vocab=60)     # each rank its own scene
        if a.graph and dev.type == "cuda":
            tr.step(data, next_data=data)
            assert tr.enable_graph(data, warmup=1), tr.graph_error
            for _ in range(a.steps):
                tr.step(data, next_data=data)
            tr.check_health()
            if a.overlap:
                assert tr.boundary_launches >= 3, tr.boundary_launches   # eager step, warm-up, capture: the overlapped tail ran
                # every rank's bucket holds the same reduced gradient, bit for bit (both slices travelled)
                torch.cuda.synchronize()
                red = [torch.empty_like(tr.bucket.flat) for _ in range(world)]
                dist.all_gather(red, tr.bucket.flat.clone())
                assert all(torch.equal(red[0], r) for r in red), "reduced buckets differ between ranks"
        else:
            tr._setup(dict(data))
            for _ in range(a.steps):
                tr._core(dict(data), with_optimizer=False)
                tr.bucket.pack()
                local = tr.bucket.flat.clone()
                gathered = [torch.empty_like(local) for _ in range(world)]
                dist.all_gather(gathered, local)
                tr._optimizer_step(None)                     # pack (no-op now) + all-reduce mean + Adam
                want = torch.stack(gathered).sum(0) / world
                got = tr.bucket.flat * tr.grad_scale      # (FlatAdam folds 1 / world into its own pass: the bucket holds the sum)
                assert torch.allclose(got, want, rtol=1e-6, atol=1e-9), float((got - want).abs().max())
                red = [torch.empty_like(got) for _ in range(world)]
                dist.all_gather(red, got.clone())
                assert all(torch.equal(red[0], r) for r in red), "reduced buckets differ between ranks"
        flat_p = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        allp = [torch.empty_like(flat_p) for _ in range(world)]
        dist.all_gather(allp, flat_p)
        assert all(torch.equal(allp[0], q) for q in allp), "parameters diverged between ranks"
        assert bool(torch.isfinite(flat_p).all())
    print(f"DIST_OK rank={rank} world={world} device={dev} params={flat_p.numel()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's own rank launcher, CPU side: `python bench.py --gpus N` with no WORLD_SIZE starts N rank processes and
reports a failing rank through its exit code (here every rank fails: no GPU, and the product path has no CPU fallback)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-container check; tests/test_engine_gpu.py covers the GPU side")
def test_self_launch_reports_failing_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "rank(s) failed first" in out.stderr and ("(0, 1)" in out.stderr or "(1, 1)" in out.stderr)
    assert "---- rank 0 (exit code" in out.stderr and "---- rank 1 (exit code" in out.stderr   # every rank's last lines
    assert out.stderr.count("bench.py needs a GPU") >= 2      # (passed through live + repeated in the failing rank's report)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-container check; tests/test_engine_gpu.py covers the GPU side")
def test_one_rank_dying_at_start_ends_the_job_within_seconds():
    """Rank 1 exits before joining the process group (test knob SPACAP_BENCH_FAIL_RANK); rank 0 then sits in
    init_process_group, where the store's own timeout is 10 - 30 minutes.  The launcher polls every rank, terminates the
    survivor and reports -- the driver's `--gpus 8` run cannot burn its budget on one bad rank."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SPACAP_BENCH_FAIL_RANK="1", SPACAP_DIST_BACKEND="gloo")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert out.returncode != 0
    assert "rank(s) failed first (rank, exit code): [(1, 1)]" in out.stderr, out.stderr[-2000:]
    assert "injected failure of this rank" in out.stderr
    assert "---- rank 0 (exit code -15)" in out.stderr or "---- rank 0 (exit code -9)" in out.stderr, out.stderr[-2000:]
    assert took < 60, took      # (import torch x 2 dominates; the store timeout would be >= 600 s)

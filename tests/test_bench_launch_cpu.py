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
    assert "rank(s) failed" in out.stderr and "(0, 1)" in out.stderr and "(1, 1)" in out.stderr
    assert out.stderr.count("bench.py needs a GPU") == 2      # two separate rank processes ran
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]

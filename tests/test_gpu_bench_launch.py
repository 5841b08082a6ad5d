"""bench.py started as the driver starts it, `python bench.py --gpus N`, must bring up N ranks by itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_gpus_2_launches_two_ranks():
    """Two ranks on the one GPU of a test box (STARK_BENCH_SHARE_GPU=1: gloo carries the collectives, the numbers
    mean nothing): the line says n_gpus 2 and carries the measurements only the N > 1 branch produces."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STARK_BENCH_SHARE_GPU"] = "1"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu"], env=env, capture_output=True, text=True, timeout=840)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2
    assert line["config"]["parallelism"] == "independent x2"
    extra = line["extra"]
    assert "error" not in extra, extra.get("error")
    assert "fourstep_ntt_2^24_ms" in extra and "fri_commit_2^24_sharded_ms" in extra
    assert extra["stark_prove_2^20_sharded_ms"] > 0
    assert extra["stark_prove_2^20_sharded_proof_sha256"] == extra["stark_proof_2^20_sha256"]     # same bytes as one GPU
    assert extra["collective"]["world"] == 2 and extra["collective"]["backend"] == "gloo"
    assert [d["rank"] for d in extra["collective"]["devices"]] == [0, 1]

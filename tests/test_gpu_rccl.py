"""-m gpu: the RCCL code path of the training config (one flat-bucket gradient all-reduce + the rank-ordered
token all-gather of the codebook EMA update) executed on the one GPU the test box has: a FRESH child job -- one rank under
torch.distributed.run, collectives forced although the group has a single member -- so that dist.bring_up's nccl branch, the
all-reduce-of-ones probe, the 34 MB gradient all-reduce and the token all-gather all really run on RCCL."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_rccl_path_runs_in_a_one_rank_group(dev):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MVQ_BENCH_ONE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "1",
           str(ROOT / "bench.py"), "--gpus", "1", "--batch", "8", "--workload", "train", "--force-collectives", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["config"]["collective_backend"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    phases = line["train_phases_ms_per_step"]
    for k in ("grad_allreduce", "ema_token_allgather", "ema_assign_and_update"):
        assert k in phases and phases[k] > 0, phases
    assert line["output_finite"] and line["build_flags"] == 0 and line["plan_overrides"] == []

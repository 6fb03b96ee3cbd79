"""bench.py's N>1 control flow on a single-GPU box: `--gpus 2` spawns two ranks itself (they share
cuda:0, gloo collectives), and the gathered top-k equals the unsharded search."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "20000", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-clustered", "--no-traffic"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_spawns_two_ranks_and_gather_matches_unsharded():
    import torch
    out = _run(["--gpus", "2", "--verify-gather"])
    assert out["n_gpus"] == 2
    assert out["gather_verified"] is True
    # two visible devices: one rank per GPU over RCCL; one device (this tier's box): the ranks share it over gloo --
    # the line says which, and either way rank 1's index arrived through shard.replicate_index (engine per rank)
    want = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert out["collective_backend"] == want and (out["rccl_world"] == 2) == (want == "nccl")
    assert out["index_replication"]["bytes_per_replica"] > 20000 * 128 * 4
    assert out["index_replication"]["replicas_answer_identically"] is True
    assert len(out["per_rank"]["qps"]) == 2 and out["per_rank"]["ms_per_step_max"] >= out["per_rank"]["ms_per_step_min"]
    assert out["topk_exchange"]["allgather_us"] > 0 and out["topk_exchange"]["d2h_to_pinned_us"] > 0
    assert out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"]["bound"] == "hbm"
    # first-contact evidence for a real multi-GPU node: every rank's own roofline fraction and what its device reaches
    assert len(out["per_rank"]["roofline_frac"]) == 2 and all(f > 0 for f in out["per_rank"]["roofline_frac"])
    assert len(out["per_rank"]["peer_access"]) == 2


def test_bench_dry_run_proves_the_replication_path_in_seconds():
    """`--dry`: stop after the index is on every rank and the replicas answered 64 queries identically"""
    out = _run(["--gpus", "2", "--dry"])
    assert out["dry"] is True and out["n_gpus"] == 2
    assert out["index_replication"]["replicas_answer_identically"] is True
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["nodes"] == 20000 for r in out["ranks"])


def test_bench_single_rank_line_has_the_contract_fields():
    out = _run(["--gpus", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["n_gpus"] == 1 and out["dtype"] == "f32" and "workload" in out["config"]
    # what the roofline's denominator is, the literal one-launch-at-a-time shape, and the library's own pipelining
    rl = out["roofline"]
    assert rl["peak_hbm_spec"] == 8000.0 and rl["hbm_measured_copy"] < rl["peak_hbm_spec"]
    assert rl["lone_launch_1024"]["launches_in_flight"] == 1 and rl["lone_launch_1024"]["kernel_ms"] > 0
    assert out["host_buffers"]["batch"] == 8192 and out["host_buffers"]["value"] > 0
    assert [c["batch"] for c in out["device_call"]] == [4096, 8192, 16384]
    assert out["config"]["pipeline"]["lanes"] == 3

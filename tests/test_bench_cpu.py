"""CPU: bench.py's launch path.  `python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself
(before anything touches a GPU), rendezvous on 127.0.0.1, run the barrier / timed loop / pose gather /
max-over-ranks reduction and have rank 0 print ONE JSON line with n_gpus == 2.  `--dry-run` swaps the model
step for a host stub and RCCL for gloo (no GPU here); everything around the step is the code the driver runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_bench_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["sharding"].startswith("dp2")
    assert abs(res["value"] - 2 * 4 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]  # whole-job aggregate
    # per-rank step times and the two collectives of the job (weights broadcast, pose gather) are in the line
    assert len(res["per_rank_ms_per_step"]) == 2 and all(0 < v <= res["ms_per_step"] * 1.001 for v in res["per_rank_ms_per_step"])
    c = res["collectives"]
    assert c["weights_broadcast_ms"] >= 0 and len(c["poses_gather_ms_per_rank"]) == 2


def test_bench_single_rank_dry_run_and_world_mismatch():
    r = _run(["--steps", "2", "--warmup", "0", "--batch", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # a launcher-provided WORLD_SIZE that disagrees with --gpus is refused (never a 1-rank number labelled 8 GPUs)
    r = _run(["--gpus", "8", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr

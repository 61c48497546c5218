"""bench.py --gpus N must really run N ranks (the flag used to be ignored)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_2_started_plainly_spawns_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_counted"] == 2 and line["backend"] in ("gloo", "nccl")
    # the shard exchange ran between the two ranks on a small ragged batch and reassembled it in pair order (the keys bench.py --gpus N prints on hardware)
    ex = line["exchange"]
    assert ex["ranks"] == 2 and ex["pairs"] == 3000 and ex["gathered_identical_to_rank0_whole_batch"] is True
    assert len(ex["bounds"]) == 3 and 0 < ex["bounds"][1] < 3000 and ex["round_trip_ms"] > 0 and ex["scatter_ms"] > 0 and ex["gather_ms"] > 0


def test_world_size_that_differs_from_gpus_is_refused():
    r = _run(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in r.stderr

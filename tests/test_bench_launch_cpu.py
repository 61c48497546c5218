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


import pytest


@pytest.mark.parametrize("scenario", ["large"] + (["huge"] if os.environ.get("BSA_TEST_HUGE") else []))
def test_c_level_exchange_at_c2_pair_count_with_offsets_beyond_4_gib(tmp_path, scenario):
    """the C-level exchange (bsa_shard_scatter / bsa_shard_gather behind the RCCL-shaped wire interface, here over shared
    memory between two processes) on 100 000 pairs whose offsets in the root's blob pass 4 GiB: ranges balanced by cells, every rank's shard
    found at the offsets the scatter reports (marked pairs where they belong, nothing else in the shard), records and CIGAR words of all
    pairs back in order.  BSA_TEST_HUGE=1 adds the same with 21-23 kbp sequences: 4.4 GB through the wire, every rank's OWN shard beyond
    ... 2 GiB and the root's dense blob beyond 4 GiB (two minutes; run once in round 6, green)"""
    import test_shard_cpu as T
    got = T._run_c_exchange(2, scenario, tmp_path)
    assert all(kind == "ok" and val[0] for _, kind, val in got), got
    ranges = sorted((val[1], val[2]) for _, _, val in got)
    assert ranges[0][0] == 0 and ranges[0][1] + ranges[1][1] == 100000 and ranges[1][0] == ranges[0][1]
    assert abs(ranges[0][1] - 50000) < 1500                  # (balanced by cells: lengths are uniform, so close to the middle)

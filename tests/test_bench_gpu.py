"""GPU: bench.py keeps its output contract -- ONE JSON line on stdout with the keys the driver reads, the roofline and
cpu_baseline objects, and a parity spot check -- on a small instance of every pairwise workload."""
import json
import os
import subprocess
import sys

import pytest

import support as S

pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


@pytest.mark.parametrize("args", [
    ["--pairs", "2048", "--length", "2000", "--cpu-pairs", "20"],
    ["--workload", "edit", "--pairs", "1024", "--length", "5000", "--cpu-pairs", "10"],
    ["--workload", "edit", "--mode", "extend", "--bw", "-1", "--pairs", "512", "--length", "3000", "--cpu-pairs", "10"],
    ["--workload", "poa", "--pairs", "256", "--length", "800", "--cpu-pairs", "-1"],
    ["--workload", "poa", "--poa-source", "fixture", "--pairs", "64", "--cpu-pairs", "-1"],
], ids=["align8", "edit", "edit-extend-full", "poa", "poa-fixture"])
def test_one_json_line_with_the_contract_keys(args):
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "bench.py"), "--steps", "2", "--warmup", "1"] + args,
                       capture_output=True, text=True, timeout=900, cwd=S.ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in KEYS:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["vs_baseline"] is None and "workload" in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    if "checks" in j and "oracle_identical_first8" in j["checks"]:
        assert j["checks"]["oracle_identical_first8"] is True and j["checks"]["pairs_flagged"] == 0
    if "fixture" in args:          # the committed programs: runs with oracle/_ref absent, every best end cell the reference's
        assert j["checks"]["best_end_cell_identical_all_programs"] is True and j["checks"]["programs"] == 64 * 6 and j["cpu_baseline"]["value"] is None
    if "--workload" not in args:
        # the pairwise line says what sits outside the timed region and validates the shard exchange on its own batch
        cfg, ex = j["config"], j["exchange"]
        assert cfg["plan_ms"] > 0 and cfg["pcie_inclusive_ms"] > j["ms_per_step"] and cfg["per_rank_gcups"] == [j["value"]] and cfg["ranks_counted"] == 1
        assert ex["pairs"] == 2048 and ex["ranks"] == 1 and ex["gathered_identical_to_rank0_whole_batch"] is True and ex["host_pointer_call_identical"] is True
        assert ex["round_trip_ms"] > 0 and ex["backend"] == "nccl"
    if args[-1] != "-1":
        cb = j["cpu_baseline"]
        # one pinned process per physical core the container may use, the one-core figure beside it (poa: one core)
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["sample"]
        assert cb["cores"] == 1 or (cb["one_core_alone"] > 0 and cb["per_core"] > 0)


def test_default_run_carries_the_three_workloads():
    """`python bench.py` (what the driver runs, here with fewer steps): the headline line with "secondary" = the edit (C3) and POA lines measured in the
    same run, each with its own roofline and cpu_baseline; plan cost and the PCIe-inclusive time in config"""
    r = subprocess.run([sys.executable, os.path.join(S.ROOT, "bench.py"), "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=1500, cwd=S.ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["config"]["pairs_per_gpu"] == 100000 and j["config"]["plan_ms"] > 0 and j["config"]["pcie_inclusive_ms"] > j["ms_per_step"]
    assert j["exchange"]["gathered_identical_to_rank0_whole_batch"] is True and j["exchange"]["pairs"] == 100000
    sec = j["secondary"]
    assert len(sec) == 3 and all("error" not in d for d in sec), sec
    ed, po, wq = sec
    # round 6: whole-query bands with scores outside the static guard on the checked systolic kernel, beside its own CPU baseline (>= 300 GCUPS, >= 5 x)
    assert "bandwidth 0" in wq["config"]["workload"] and "CHK" in wq["roofline"]["kernel"] and wq["checks"]["oracle_identical_first8"] is True
    assert wq["value"] >= 300 and wq["value"] >= 5 * wq["cpu_baseline"]["value"] and wq["checks"]["pairs_flagged"] <= 20
    assert ed["metric"].startswith("GCUPS") and "edit" in ed["config"]["workload"] and ed["value"] > 0 and ed["roofline"]["frac"] > 0 and ed["cpu_baseline"]["value"] > 0
    assert ed["checks"]["oracle_identical_first8"] is True and ed["config"]["plan_ms"] > 0
    assert "poa" in po["config"]["workload"] and po["value"] > 0 and po["roofline"]["frac"] > 0 and po["checks"]["best_end_cell_identical_all_programs"] is True
    print("\n[bench.py default run] align8 %.0f GCUPS (%.1f ms/step, plan %.0f ms, PCIe-inclusive %.0f ms), edit %.0f GCUPS, poa %.1f GCUPS, whole query out of guard %.0f GCUPS; secondaries took %.0f + %.0f + %.0f s"
          % (j["value"], j["ms_per_step"], j["config"]["plan_ms"], j["config"]["pcie_inclusive_ms"], ed["value"], po["value"], wq["value"], ed["wall_s"], po["wall_s"], wq["wall_s"]))

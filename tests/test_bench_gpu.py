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
    if args[-1] != "-1":
        cb = j["cpu_baseline"]
        # one pinned process per physical core the container may use, the one-core figure beside it (poa: one core)
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["sample"]
        assert cb["cores"] == 1 or (cb["one_core_alone"] > 0 and cb["per_core"] > 0)

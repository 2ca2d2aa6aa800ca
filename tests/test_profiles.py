"""The committed round-6 profiles belong to the code in the tree.

The round-5 review found profiles measured at four different source states.  Every `profiles/r06_*` summary names the hash of
`femto_amd/csrc` it was measured at (`benchlib.common.source_hash`, written by `tools/profile_round.sh` and the round scripts);
this test fails when a kernel or host source changes without the rounds being measured again
(`tools/final_round_r06.sh`, `tools/budget_sweep*.sh`, then `python tools/promote_profiles.py <tags>`)."""
import glob
import json
import os
import re

from benchlib.common import source_hash

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HASH_AT = re.compile(r"hash[^\n]*?\b([0-9a-f]{16})\b")


def _summaries():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*.txt")) + glob.glob(os.path.join(ROOT, "profiles", "r06_*.md")))


def test_round_profiles_are_at_the_tree_s_source_hash():
    here = source_hash()
    files = _summaries()
    assert len(files) >= 15, files
    for f in files:
        found = set(HASH_AT.findall(open(f).read()))
        assert found == {here}, (os.path.basename(f), found, here)


def test_round_bench_lines_carry_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*_bench.json")))
    assert len(lines) >= 10
    for f in lines:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
            assert k in d, (os.path.basename(f), k)
        r = d["roofline"]
        if f.endswith("r06_mode0_bench.json"):      # north_star's literal kernel, 2 steps of 226 ms, run with --cpu-sample 0: its figures
            assert r is None                         # are the mode0_* scalars of the default line's roofline and r06_mode0_stats.txt
            continue
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0, os.path.basename(f)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert "workload" in d["config"] and "model" not in d["config"]
        # the rocprofv3 --kernel-trace --stats summary of the same command sits beside the line
        stats = f.replace("_bench.json", "_stats.txt")
        assert os.path.exists(stats), stats
    d = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r06_default_bench.json")) if l.startswith("{")][-1])
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["vs_baseline"] is None and d["steps"] >= 200

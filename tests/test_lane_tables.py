"""CPU check of the DERIVED device tables (128-byte segment lines, cum/hint, RLE skip tables, child
links, Occ bases) the loader builds: a test-only host emulation of the lane kernels' table walk
(tests/emul/lane_emul.cpp, bounds-checked) must reproduce the reference's per-row golden vectors."""
import os
import subprocess

import numpy as np
import pytest

from conftest import INDEX_FIXTURES, ROOT


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "lane_emul")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "emul", "lane_emul.cpp"),
                    os.path.join(ROOT, "femto_amd", "csrc", "host_index.cpp")], check=True)
    return exe


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_lane_tables_reproduce_reference_rows(fixtures, emul, tmp_path, name):
    fx = fixtures(name)
    out = str(tmp_path / "rows.bin")
    for path in [fx.index] + ([fx.flat] if os.path.exists(fx.flat) else []):
        subprocess.run([emul, path, out], check=True)
        rows = np.fromfile(out, dtype=np.dtype([("ch", "<u2"), ("occ", "<i8"), ("off", "<i8")]))
        g = fx.gold
        assert np.array_equal(rows["ch"], g["L"])
        assert np.array_equal(rows["off"], g["off"])
        # occ = C[ch] + block_occs[ch][block] + occs_in_block
        block_size = int(fx.params.split("block_size=")[1].split(",")[0]) if "block_size=" in fx.params else 1 << 27
        blk = np.arange(len(g["L"])) // block_size
        want = g["C"][g["L"]] + g["block_occs"][g["L"], blk] + g["occ"]
        assert np.array_equal(rows["occ"], want)

"""The drop-in, compiled and run for real.

integration/femto_amd_shim.c defines femto's parallel_count / parallel_locate / parallel_locate_range
(src/main/femto_internal.h:63-75) over the C ABI of include/femto_amd.h.  oracle/Makefile compiles it against the
reference's headers and links the reference's OWN callers with it:
  * ref_tool_amd   -- oracle/ref_tool.c, whose count / locate commands make the calls of query_tool.c:133-206;
  * index_test_amd -- the reference's integration test src/main/index_test.c, compiled where it lies.
CPU: the shim compiles (-Wall -Werror) and the binaries link with the four functions coming from the shim.
GPU: the reference binaries answer from the GPU, bit-exact against the goldens / their own assertions.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import INDEX_FIXTURES, ROOT
from oracle import pyoracle as po

HAVE_REFERENCE = os.path.isdir("/root/reference/src/main")


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference headers (build container only)")
def test_shim_compiles_and_links_against_reference():
    import femto_amd
    femto_amd.lib()     # the product library must be built first: the shim links against it
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "shim"], check=True)
    for exe in (po.REF_TOOL_AMD, po.INDEX_TEST_AMD):
        assert os.path.exists(exe)
    so = os.path.join(po.REF_DIR, "libfemto_ref_amd.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], check=True, stdout=subprocess.PIPE).stdout.decode().split("\n")
    defined = {ln.split()[-1] for ln in syms if ln.strip()}
    # the batch entry points come from the shim, the reference's own bodies stay linked under femto_cpu_*
    for name in ("parallel_count", "parallel_locate", "parallel_locate_range", "femto_stop_server",
                 "femto_cpu_parallel_count", "femto_cpu_parallel_locate", "femto_cpu_parallel_locate_range",
                 "femto_cpu_stop_server", "femto_amd_shim_forget"):
        assert name in defined, name
    und = subprocess.run(["nm", "-D", "--undefined-only", so], check=True, stdout=subprocess.PIPE).stdout.decode()
    for name in ("femto_amd_open", "femto_amd_parallel_count", "femto_amd_parallel_locate", "femto_amd_parallel_locate_range"):
        assert name in und, name


def _have_amd_tools():
    return os.path.exists(po.REF_TOOL_AMD) and os.path.exists(po.INDEX_TEST_AMD)


@pytest.mark.gpu
@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_reference_caller_answers_from_gpu(fixtures, tmp_path, name):
    """ref_tool's count / locate commands call parallel_count / parallel_locate exactly as the reference's tools do;
    linked with the shim the answers come from the GPU and must equal the goldens captured from the CPU reference."""
    if not _have_amd_tools():
        pytest.skip("oracle/_ref/ref_tool_amd was not prebuilt (needs /root/reference at build time)")
    fx = fixtures(name)
    plen, flat, starts = fx.patterns
    pf = str(tmp_path / "p.fpat")
    po.write_fpat_flat(pf, plen, flat)
    out = str(tmp_path / "count.bin")
    subprocess.run([po.REF_TOOL_AMD, "count", fx.index, pf, out], check=True, timeout=300)
    r = np.fromfile(out, dtype=np.int64)
    n = len(plen)
    assert np.array_equal(r[:n], fx.gold["count_first"]) and np.array_equal(r[n:], fx.gold["count_last"])
    for mo, g_noccs, g_offs in fx.locate_cases():
        out = str(tmp_path / f"loc{mo}.bin")
        subprocess.run([po.REF_TOOL_AMD, "locate", fx.index, pf, str(mo), out], check=True, timeout=300)
        raw = np.fromfile(out, dtype=np.uint8)
        noccs = raw[:4 * n].view(np.int32)
        offs = raw[4 * n:].view(np.int64)
        assert np.array_equal(noccs, g_noccs), mo
        assert np.array_equal(offs, g_offs), mo
    # the flattened container goes through the same path translator entry (isfile)
    flat_path = fx.flat
    if not os.path.exists(flat_path):      # only one fixture ships the reference's own flattened file
        import femto_amd
        flat_path = str(tmp_path / "index.flat")
        femto_amd.flatten_index(fx.index, flat_path)
    out = str(tmp_path / "count_flat.bin")
    subprocess.run([po.REF_TOOL_AMD, "count", flat_path, pf, out], check=True, timeout=300)
    r = np.fromfile(out, dtype=np.int64)
    assert np.array_equal(r[:n], fx.gold["count_first"]) and np.array_equal(r[n:], fx.gold["count_last"])


@pytest.mark.gpu
def test_reference_index_test_passes_on_gpu(tmp_path):
    """src/main/index_test.c (test_construct + make_and_test_index over 6 texts x 4 parameter sets): every
    parallel_count is compared with a brute-force count and every parallel_locate offset with the text by the
    reference's own assertions (index_test.c:351-434) -- here with both calls served by the GPU."""
    if not _have_amd_tools():
        pytest.skip("oracle/_ref/index_test_amd was not prebuilt (needs /root/reference at build time)")
    r = subprocess.run([po.INDEX_TEST_AMD], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = r.stdout.decode(errors="replace")[-2000:]
    assert r.returncode == 0, tail
    assert "All index tests PASSED" in tail, tail

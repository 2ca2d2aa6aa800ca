"""The drop-in, compiled and run for real.

integration/femto_amd_shim.c defines femto's parallel_count / parallel_locate / serial_locate / parallel_locate_range
(src/main/femto_internal.h:63-75) over the C ABI of include/femto_amd.h.  oracle/Makefile compiles it against the
reference's headers and links the reference's OWN callers with it:
  * ref_tool_amd   -- oracle/ref_tool.c, whose count / locate commands make the calls of query_tool.c:133-206;
  * index_test_amd -- the reference's integration test src/main/index_test.c, compiled where it lies.
CPU: the shim compiles (-Wall -Werror) and the binaries link with the five functions coming from the shim.
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
    for name in ("parallel_count", "parallel_locate", "serial_locate", "parallel_locate_range", "femto_stop_server",
                 "femto_cpu_parallel_count", "femto_cpu_parallel_locate", "femto_cpu_serial_locate", "femto_cpu_parallel_locate_range",
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
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_serial_locate_answers_from_gpu(fixtures, tmp_path, name):
    """serial_locate (src/main/femto.c:402, called by query_tool.c:157) has a clamp of its own -- 1+last-first >= max_occs --
    so it is compared with the genuine CPU reference run beside it (oracle/_ref/ref_tool serial) on every fixture pattern,
    at limits that put patterns exactly at, one above and far above the clamp."""
    if not _have_amd_tools() or not po.have_ref():
        pytest.skip("oracle/_ref was not prebuilt (needs /root/reference at build time)")
    fx = fixtures(name)
    plen, flat, starts = fx.patterns
    counts = np.maximum(fx.gold["count_last"] - fx.gold["count_first"] + 1, 0)
    # only patterns that occur: the reference's serial_locate cleans up a query it never set up when a pattern has no rows
    # (femto.c:410-476: `ctx` is malloc()ed, setup_locate_range runs per located row, cleanup_locate_range always) -- an
    # assertion failure in cleanup_parallel_query (server.c:3962), so its CPU twin cannot be run on them
    keep = np.nonzero(counts > 0)[0]
    flat = np.concatenate([flat[starts[i]:starts[i] + plen[i]] for i in keep]).astype(np.uint16)
    plen, counts = plen[keep], counts[keep]
    pf = str(tmp_path / "p.fpat")
    po.write_fpat_flat(pf, plen, flat)
    multi = sorted(set(int(c) for c in counts if 1 < c < 200))
    limits = [1, 7] + ([multi[0], multi[0] - 1] if multi else [])
    for mo in limits:
        a, b = str(tmp_path / f"cpu{mo}.bin"), str(tmp_path / f"gpu{mo}.bin")
        subprocess.run([po.REF_TOOL, "serial", fx.index, pf, str(mo), a], check=True, timeout=600)
        subprocess.run([po.REF_TOOL_AMD, "serial", fx.index, pf, str(mo), b], check=True, timeout=600)
        ra, rb = np.fromfile(a, dtype=np.uint8), np.fromfile(b, dtype=np.uint8)
        assert np.array_equal(ra, rb), (name, mo)
        n = len(plen)
        assert np.array_equal(rb[:4 * n].view(np.int32), np.minimum(counts, mo)), mo      # femto.c:427-428


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

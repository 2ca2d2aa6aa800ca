"""CPU (`-m "not gpu"`) tests of the product's host side: C-ABI surface, index loader, index writer."""
import ctypes as C
import filecmp
import os
import re
import shutil

import numpy as np
import pytest

import femto_amd
from conftest import GOLDEN, INDEX_FIXTURES, ROOT, WRITER_FIXTURES
from sa_util import suffix_array


def test_cabi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "femto_amd.h")).read()
    names = sorted(set(re.findall(r"\b(femto_amd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = femto_amd.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_loader_parses_reference_built_indexes(fixtures, name):
    fx = fixtures(name)
    ix = femto_amd.Index(fx.index, device=-1)      # parse + validate only: no GPU here
    g = fx.gold
    assert ix.info.total_length == len(g["L"])
    assert ix.info.number_of_blocks == g["block_occs"].shape[1]
    assert ix.info.number_of_documents == len(fx.docs)
    assert ix.info.text_size_bits == max(1, int(ix.info.total_length).bit_length())
    assert ix.info.total_buckets == -(-ix.info.total_length // ix.info.bucket_size)
    if name in WRITER_FIXTURES:
        assert ix.info.chunk_size == -1            # index_documents(map=NULL), construct.c:604
    else:
        assert ix.info.chunk_size == 256
    # resolve_location (index.c:1587): doc_ends are cumulative doc_len+1
    ends = np.cumsum([len(d) + 1 for d in fx.docs])
    for off in [0, int(ends[0]) - 1, int(ends[-1]) - 1] + ([int(ends[0])] if len(ends) > 1 else []):
        d = int(np.searchsorted(ends, off, side="right"))
        assert ix.resolve_location(off) == (d, off - (int(ends[d - 1]) if d else 0))


def test_flattened_and_directory_forms_parse_alike(fixtures):
    fx = fixtures("acgt48k")
    a = femto_amd.Index(fx.index, device=-1)
    b = femto_amd.Index(fx.flat, device=-1)
    for f, _ in femto_amd.Info._fields_:
        if f != "image_bytes":
            assert getattr(a.info, f) == getattr(b.info, f), f


def test_loader_error_behaviour(fixtures, tmp_path):
    fx = fixtures("counter400_default")
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        femto_amd.Index(str(tmp_path / "nope"), device=-1)
    assert e.value.code == 2                        # ERR_IO
    bad = tmp_path / "bad"
    shutil.copytree(fx.index, bad)
    raw = bytearray(open(bad / "01", "rb").read())
    raw[0] ^= 0xFF                                  # DATA_BLOCK_START magic
    open(bad / "01", "wb").write(raw)
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        femto_amd.Index(str(bad), device=-1)
    assert e.value.code == 4 and "block start" in str(e.value)   # ERR_FORMAT "Invalid block start" (index.c:1361)
    bad2 = tmp_path / "bad2"
    shutil.copytree(fx.index, bad2)
    raw = bytearray(open(bad2 / "00", "rb").read())
    raw[79] ^= 1                                    # wavelet settings number
    open(bad2 / "00", "wb").write(raw)
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        femto_amd.Index(str(bad2), device=-1)
    assert e.value.code == 4
    bad3 = tmp_path / "bad3"
    shutil.copytree(fx.index, bad3)
    raw = open(bad3 / "01", "rb").read()
    open(bad3 / "01", "wb").write(raw[: len(raw) // 2])       # truncated data block
    with pytest.raises(femto_amd.FemtoAmdError):
        femto_amd.Index(str(bad3), device=-1)


def test_no_cpu_fallback(fixtures):
    """The product must fail loudly without a GPU: no silent host path exists."""
    import torch
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=-1)
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        ix.count([np.array([70, 72], dtype=np.uint16)])
    assert e.value.code == 6
    with pytest.raises(femto_amd.FemtoAmdError):
        ix.locate([np.array([70], dtype=np.uint16)], 3)
    if not torch.cuda.is_available():
        with pytest.raises(femto_amd.FemtoAmdError) as e:
            femto_amd.Index(fx.index, device=0)
        assert e.value.code == 6 and "no CPU fallback" in str(e.value)
        with pytest.raises(femto_amd.FemtoAmdError):
            femto_amd.build_index(str(fx.dir) + "/x", [b"ACGT"], device=0)


def test_options_struct(fixtures):
    """femto_amd_options_t: init = every field auto; a struct of another size is refused; a parse-only handle opens with options"""
    import ctypes as C
    o = femto_amd.Options()
    assert o.struct_size == C.sizeof(femto_amd.Options) and o.rank_mode == -1 and o.hbm_budget_bytes == -1 and o.host_d2h_staged == -1
    o = femto_amd.Options(level_table_syms=8, hbm_budget_bytes=1 << 30, context2_table=0)
    assert (o.level_table_syms, o.hbm_budget_bytes, o.context2_table, o.text) == (8, 1 << 30, 0, -1)
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=-1, options=o)
    assert ix.info.total_length == len(fx.prepared_text())
    ix.close()
    o.struct_size = 12
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        femto_amd.Index(fx.index, device=-1, options=o)
    assert e.value.code == 3
    with pytest.raises(KeyError):
        femto_amd.Options(no_such_field=1)


def test_search_cli_output_formats():
    """femto_search's text and JSON output (src/main_cc/search_tool.cc:889-894, :1038-1048, :1075-1087, :1105-1114, print_matches
    :470-519; alphatos src/main/index_types.h:104-118; encode_ch_json src/main/json.c:35-62) restated as a table: format
    string -> expected bytes for a golden result.  The reference tool itself needs flex/bison + RE2 and cannot be built
    here, so its formats are RESTATED (from the printf calls cited per row), not diffed against its output; femto_amd_search
    prints through the same table (--formats) and renders a fixed result with its own print functions (--format-selftest)."""
    import subprocess
    from femto_amd import build as b
    b.build_tools()
    # (name, the reference's format string with PRIi64 = "li", source line)
    ref = [("matches_row_head", '% 4li "', "1084"), ("matches_row_tail", '"%c', "1086"), ("total", "% 4li total matches%c", "1112"),
           ("doc_info", "%.*s", "478"), ("doc_sep", "%c%s", "477"), ("offsets_lead", "%c\\t", "480"), ("offset", " %li", "499"), ("list_end", "%c", "519"),
           ("by_index_head", "Results from %s\\n", "1038"), ("by_index_row", '% 4li [%li,%li] "', "1046"),
           ("json_open", '{\\n "pattern":"', "890-891"), ("json_results", '",\\n "results":[\\n   ', "893"), ("json_row_sep", ",\\n   ", "1077"),
           ("json_total", ',\\n "total":%li', "1111"), ("json_close", "\\n}\\n", "1114")]
    rows = [r.split("\t") for r in subprocess.run([b.SEARCH, "--formats"], capture_output=True, check=True).stdout.decode().splitlines()]
    table = {r[0]: (r[1], r[2]) for r in rows}
    assert set(table) == {n for n, _, _ in ref}
    for name, fmt, line in ref:
        assert table[name] == (fmt, f"search_tool.cc:{line}"), name
    f = {n: fm.replace("li", "d").replace("\\t", "\t").replace("\\n", "\n") for n, fm, _ in ref}      # Python's % renders C's "% 4li" as "% 4d"

    def docs(lst, offsets, sep):
        out, first = b"", True
        for info, offs in lst:
            if not first:
                out += (f["doc_sep"] % (sep, "")).encode()
            first = False
            out += info.encode()                           # "%.*s"
            if offsets:
                out += (f["offsets_lead"] % sep).encode() + b"".join((f["offset"] % o).encode() for o in offs)
        return out + ((f["list_end"] % sep).encode() if lst else b"")

    golden = [("doc0.txt", [3, 17, 4242]), ("dir/doc1", [0])]
    for sep, flag in (("\n", []), ("\0", ["--null"])):
        # the matched strings go through alphatos: backslash and quote escaped, other bytes printable or \xNN, codes below the
        # bytes \x-NN (alpha 2 = 5 - 3 is the end-of-document marker)
        want = ((f["matches_row_head"] % 7).encode() + b"the" + (f["matches_row_tail"] % sep).encode()
                + (f["matches_row_head"] % 12345).encode() + b'a \\"b\\"\\\\\\x01\\xff' + (f["matches_row_tail"] % sep).encode()
                + (f["matches_row_head"] % 1).encode() + b"\\x-03A" + (f["matches_row_tail"] % sep).encode()
                + (f["total"] % (12352, sep)).encode() + docs(golden, True, sep) + docs(golden, False, sep) + docs([], True, sep)
                + (f["total"] % (0, sep)).encode())
        # JSON (print_matches' json branches): a new document closes the one before it with "] ]," -- the offsets list of the
        # LAST document of an index is closed by " ] ] "; '|' in an info string separates grouped documents
        want += (b'[ ["a","b"], [3, 17] ],\n   [ ["c\\"d"], [0 ] ] ' + b',\n   [ ["e"], [ ] ] ' + b"\n"
                 + b'q\\\\\\"\\\\\\\\\\\\x01' + b"\n")
        got = subprocess.run([b.SEARCH] + flag + ["--format-selftest"], capture_output=True, check=True).stdout
        assert got == want, (sep, got, want)
    assert want.startswith(b"   7 \"the\"\x00 12345 ")      # "% 4d": at least four columns, a blank for the sign


def test_bseq_encoder_is_byte_identical_to_reference():
    """The 12 sequences x 3 segment-type modes of wtree_test.c:440-580, images captured from the
    reference's bseq_construct_forcetype."""
    kat = np.load(os.path.join(GOLDEN, "bseq_kat.npz"))
    for i in range(int(kat["nseq"])):
        raw, nbits = kat[f"s{i}_raw"], int(kat[f"s{i}_nbits"])
        for t in range(3):
            z = femto_amd.bseq_encode(raw.tobytes(), nbits, t - 1)
            assert np.array_equal(z, kat[f"s{i}_t{t}_z"]), (i, t)


@pytest.mark.parametrize("name", WRITER_FIXTURES)
def test_index_writer_is_byte_identical_to_reference(fixtures, tmp_path, name):
    """same documents + parameters -> the same bytes as index_documents(map=NULL)."""
    fx = fixtures(name)
    sa = suffix_array(fx.prepared_text())
    out = str(tmp_path / "mine")
    infos = [os.path.basename(p) for p in fx.doc_paths]
    femto_amd.build_index_from_sa(out, fx.docs, sa, params=None if fx.params == "-" else fx.params, infos=infos)
    ref_files = sorted(f for f in os.listdir(fx.index) if f != "_femto_index")
    my_files = sorted(f for f in os.listdir(out) if f != "_femto_index")
    assert ref_files == my_files
    for f in ref_files:
        assert filecmp.cmp(os.path.join(fx.index, f), os.path.join(out, f), shallow=False), f


def test_builder_parameter_validation(tmp_path):
    docs = [np.frombuffer(b"ACGTACGT", dtype=np.uint8)]
    sa = suffix_array(np.concatenate([docs[0].astype(np.uint16) + 5, [2]]))
    for bad in ("block_size=10,bucket_size=4", "bucket_size=0", "bogus=1", "mark_period=0"):
        with pytest.raises(femto_amd.FemtoAmdError):
            femto_amd.build_index_from_sa(str(tmp_path / "x"), docs, sa, params=bad)


def test_flatten_is_byte_identical_to_reference(fixtures, tmp_path):
    fx = fixtures("acgt48k")
    out = str(tmp_path / "mine.flat")
    femto_amd.flatten_index(fx.index, out)
    assert filecmp.cmp(fx.flat, out, shallow=False)
    assert femto_amd.Index(out, device=-1).info.total_length == femto_amd.Index(fx.index, device=-1).info.total_length


def test_multiquery_tool_fails_loudly_without_gpu(fixtures):
    """The C++ host program (tools/femto_amd_multiquery.cpp) is pure C ABI: without a GPU it must stop with the
    library's error, not fall back to anything."""
    import subprocess
    import torch
    from femto_amd import build as b
    tool = b.build_tools()
    fx = fixtures("acgt48k")
    if not torch.cuda.is_available():
        r = subprocess.run([tool, fx.index, "-count"], input=b"# number=1 length=4 x\nACGT", capture_output=True)
        assert r.returncode == 1 and b"no CPU fallback" in r.stderr


def test_document_info_matches_what_the_reference_stored(fixtures):
    """document_info (src/main/index.c:1768): the reference-built fixtures carry the document file names."""
    for name in ["eng2doc", "runs3doc", "chunks2doc", "acgt48k"]:
        fx = fixtures(name)
        ix = femto_amd.Index(fx.index, device=-1)
        assert ix.info.number_of_documents == len(fx.docs)
        for d, p in enumerate(fx.doc_paths):
            assert ix.document_info(d) == os.path.basename(p).encode()
        with pytest.raises(femto_amd.FemtoAmdError):
            ix.document_info(len(fx.docs))
        ix.close()


def test_damaged_tables_are_rejected_at_load(fixtures, tmp_path):
    """The kernels trust the header's cumulative tables for every row number they form: a header whose C table is not
    monotone, whose occurrence tables do not add up to the bucket sizes, or whose document ends run backwards must
    fail with ERR_FORMAT at open (tools/fuzz_gpu.py exercises the same on the GPU box)."""
    import shutil
    import struct
    fx = fixtures("eng2doc")
    src = fx.index

    def damaged(name, edit):
        dst = str(tmp_path / name)
        shutil.copytree(src, dst)
        data = bytearray(open(os.path.join(dst, "00"), "rb").read())
        edit(data)
        open(os.path.join(dst, "00"), "wb").write(data)
        return dst

    c_off = 88                                            # C[261] follows the 88-byte block header (index.c:870-898)

    def swap_c(d):                                        # two adjacent non-zero-width entries swapped -> not monotone
        vals = list(struct.unpack(">261q", d[c_off:c_off + 261 * 8]))
        k = next(i for i in range(10, 259) if vals[i] < vals[i + 1] < vals[i + 2])
        vals[k + 1], vals[k + 2] = vals[k + 2], vals[k + 1]
        d[c_off:c_off + 261 * 8] = struct.pack(">261q", *vals)

    def bump_block_occs(d):                               # one block_occs entry + 1: buckets no longer add up
        nb = femto_amd.Index(src, device=-1).info.number_of_blocks
        off = c_off + 261 * 8 + 8 * (ord("e") + 5) * nb + 8 * (nb - 1)
        v = struct.unpack(">q", d[off:off + 8])[0]
        d[off:off + 8] = struct.pack(">q", v + 1)

    for name, edit in (("c_swapped", swap_c), ("occs_bumped", bump_block_occs)):
        with pytest.raises(femto_amd.FemtoAmdError) as ei:
            femto_amd.Index(damaged(name, edit), device=-1)
        assert ei.value.code == 4, name                   # ERR_FORMAT
    femto_amd.Index(src, device=-1).close()               # the undamaged copy still opens


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/femto_amd.h is the drop-in boundary for a C code base (femto is C99): it must compile with a C compiler,
    pedantically, and a C program must link against the library and get a clean error without a GPU."""
    import subprocess
    from femto_amd import build as b
    src = tmp_path / "shim.c"
    src.write_text(r'''
#include <stdio.h>
#include "femto_amd.h"
int main(int argc, char** argv) {
  femto_amd_index_t* ix = NULL;
  int rc = femto_amd_open(argc > 1 ? argv[1] : "/nonexistent", -1, &ix);
  printf("rc=%d msg=%s\n", rc, femto_amd_last_error());
  if (rc == 0) {
    femto_amd_info_t info;
    if (femto_amd_info(ix, &info)) return 3;
    printf("rows=%lld docs=%lld\n", (long long) info.total_length, (long long) info.number_of_documents);
    {
      int64_t first = 0, last = 0;
      int plen = 1;
      const uint16_t sym = 70, *pat = &sym;
      rc = femto_amd_parallel_count(ix, 1, &plen, &pat, &first, &last);   /* parse-only handle: must refuse, not compute */
      printf("count rc=%d\n", rc);
    }
    femto_amd_close(ix);
  }
  return 0;
}
''')
    exe = tmp_path / "shim"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(b.LIB)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-o", str(exe), str(src),
                    "-L", libdir, "-lfemto_amd", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert "rc=2" in out                                       # ERR_IO for a missing path, femto's code
    fx_index = os.path.join(str(tmp_path), "ix")
    import tarfile
    with tarfile.open(os.path.join(GOLDEN, "eng2doc.tar.gz")) as tf:
        tf.extractall(fx_index)
    out = subprocess.run([str(exe), os.path.join(fx_index, "index")], capture_output=True, text=True, check=True).stdout
    assert "rc=0" in out and "docs=2" in out and "count rc=6" in out     # ERR_INVALID: no device behind this handle


def test_worker_pool_runs_every_job_on_every_worker(tmp_path):
    """host_pipeline.hpp's staging pool (spin, then sleep): 20 000 back-to-back jobs with pauses long enough for the workers
    to fall asleep in between -- every worker runs every job exactly once, none is lost across the spin / sleep hand-over."""
    src = tmp_path / "pool.cpp"
    src.write_text('''
#include "host_pipeline.hpp"
#include <cstdio>
int main() {
  const int T = 4;
  femto_amd::WorkerPool pool(T);
  std::vector<long> acc(T, 0);
  long want = 0;
  for (int rep = 0; rep < 20000; rep++) {
    pool.run([&](int t, int nt) { acc[size_t(t)] += t + rep % 3 + nt; });
    for (int t = 0; t < T; t++) want += t + rep % 3 + T;
    if (rep % 4000 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));
  }
  long total = 0;
  for (long v : acc) total += v;
  std::printf("%s\\n", total == want ? "ok" : "MISMATCH");
  return total != want;
}
''')
    exe = tmp_path / "pool"
    import subprocess
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "femto_amd", "csrc"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout


def _pack_keys_reference(dense, bits, plen, flat, starts):
    """the key format of count_keys_kernel, by definition: field j from the top = dense code of the j-th symbol from the end"""
    nsym = 63 // bits
    keys = np.zeros(len(plen), dtype=np.uint64)
    ok = True
    for i, (l, s) in enumerate(zip(plen.tolist(), starts.tolist())):
        if l > nsym:
            return None, False
        key = 0
        for j in range(l):
            sym = int(flat[s + l - 1 - j])
            c = int(dense[sym]) if sym < len(dense) else 0
            ok = ok and c != 0
            key |= c << (64 - bits * (j + 1))
        keys[i] = key
    return keys, ok


@pytest.mark.parametrize("bits,alphabet", [(3, b"ACGT"), (3, b"\x00ACGNT"), (7, bytes(range(32, 127))), (8, bytes(range(3, 256))), (1, b"A"), (2, b"AB")])
def test_host_key_packing_simd_equals_scalar_equals_definition(bits, alphabet):
    """host_pack.cpp: the AVX-512 VBMI / BMI2 packing loop of the host-pointer batches against the plain loop and against the
    key format written out in Python -- every length 0 .. 63 / bits, patterns starting at odd addresses and at the very end of
    the symbol array (the masked load must not read past it), bytes 251..255 (symbols 256..260: the scalar way inside the
    SIMD loop), a symbol without a code (the chunk is given up: return 0, same in both)."""
    lib = femto_amd.lib()
    fn = lib.femto_amd_host_pack_keys
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(bits * 1000 + len(alphabet))
    dense = np.zeros(261, dtype=np.uint8)
    chars = sorted(set(alphabet))[: (1 << bits) - 1]
    for rank, ch in enumerate(chars):
        dense[ch + 5] = rank + 1           # alpha_t = byte + 5 (index_types.h:64-69)
    nsym = 63 // bits
    lens = np.concatenate([np.arange(0, nsym + 1), rng.integers(0, nsym + 1, size=3000)]).astype(np.int32)
    starts = np.zeros(len(lens), dtype=np.int64)
    starts[1:] = np.cumsum(lens[:-1])
    total = int(lens.sum())
    flat = (np.array(chars, dtype=np.uint16)[rng.integers(0, len(chars), size=total)] + 5).astype(np.uint16)

    def run(force_scalar, flat_arr):
        keys = np.full(len(lens), 0xDEADBEEF, dtype=np.uint64)
        used = C.c_int(-1)
        rc = fn(dense.ctypes.data, 261, bits, len(lens), lens.ctypes.data, flat_arr.ctypes.data, starts.ctypes.data, force_scalar,
                keys.ctypes.data, C.byref(used))
        return rc, keys, used.value

    want, ok = _pack_keys_reference(dense, bits, lens, flat, starts)
    assert ok
    rc_s, k_s, used_s = run(1, flat)
    rc_v, k_v, used_v = run(0, flat)
    assert rc_s == 1 and rc_v == 1 and used_s == 0
    assert np.array_equal(k_s, want)
    assert np.array_equal(k_v, want)
    # a symbol without a code somewhere: both give the chunk up
    if total:
        bad = flat.copy()
        bad[total // 2] = 2 if dense[2] == 0 else 300
        assert run(1, bad)[0] == 0 and run(0, bad)[0] == 0
    # a pattern longer than a key holds
    long_lens = lens.copy()
    if nsym < 63:
        long_lens[5] = nsym + 1
        keys = np.zeros(len(lens), dtype=np.uint64)
        big = np.concatenate([flat, flat[: nsym + 2]])
        for force in (0, 1):
            assert fn(dense.ctypes.data, 261, bits, len(lens), long_lens.ctypes.data, big.ctypes.data, starts.ctypes.data, force,
                      keys.ctypes.data, None) == 0


def test_host_key_packing_never_reads_past_the_last_symbol():
    """The AVX-512 loop loads 32 symbols' worth of lanes per pattern with the lanes beyond the pattern masked off: a pattern that
    ends on the last bytes of a mapped page, with an inaccessible page behind it, must pack without a fault (and equal the scalar keys)."""
    import mmap
    lib = femto_amd.lib()
    fn = lib.femto_amd_host_pack_keys
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    m = mmap.mmap(-1, 2 * page)
    base = C.addressof(C.c_char.from_buffer(m))
    assert libc.mprotect(base + page, page, 0) == 0          # PROT_NONE behind the first page
    try:
        dense = np.zeros(261, dtype=np.uint8)
        for rank, ch in enumerate(b"ACGT"):
            dense[ch + 5] = rank + 1
        nsym_page = page // 2
        flat = np.frombuffer(m, dtype=np.uint16, count=nsym_page)      # the accessible page as symbols
        rng = np.random.default_rng(9)
        flat[:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, nsym_page)].astype(np.uint16) + 5
        lens = np.array([1, 2, 7, 8, 9, 15, 16, 17, 20, 21], dtype=np.int32)
        starts = (nsym_page - lens).astype(np.int64)                   # every pattern ends with the page
        want, ok = _pack_keys_reference(dense, 3, lens, flat, starts)
        assert ok
        for force in (1, 0):
            keys = np.zeros(len(lens), dtype=np.uint64)
            assert fn(dense.ctypes.data, 261, 3, len(lens), lens.ctypes.data, base, starts.ctypes.data, force, keys.ctypes.data, None) == 1
            assert np.array_equal(keys, want), force
        del flat
    finally:
        libc.mprotect(base + page, page, 3)
        m.close()

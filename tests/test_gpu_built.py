"""`-m gpu`: indexes BUILT on the GPU box (GPU suffix sorter + byte-identical writer) against the oracle and, for configs[0],
the genuine reference run beside it: random indexes, long patterns / text tails, context tables, the suffix sorter's paths."""
import ctypes as C
import os

import numpy as np
import pytest

import femto_amd
from conftest import INDEX_FIXTURES
from femto_amd import textgen as tg
from gpu_common import MODES, _open, _set_mode, _torchrun, assert_row_free_equals, device_locate
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _random_index(tmp_path, text, params, name):
    out = str(tmp_path / name)
    femto_amd.build_index(out, [text], params=params, infos=[name], device=0)
    return out


@pytest.mark.parametrize("mode", MODES)
def test_gpu_built_index_vs_oracle_medium(tmp_path, gpu_ok, mode):
    """4 MiB random ACGT with the reference's DEFAULT parameters (bucket 2^20 rows): the GPU
    suffix sorter + writer build the index, the HIP query path is compared with the oracle on
    100 k patterns (BASELINE config 1 shape, scaled), plus size-independent properties."""
    text = tg.t_acgt(1 << 22, 2024)
    path = _random_index(tmp_path, text, None, "acgt4m")
    ix = femto_amd.Index(path, device=0)
    assert ix.rank_mode == 3 and not ix.pack_info()["available2"]    # DNA alphabet: the packed lines are the default path
    ix.close()
    ix = _open(path, mode)
    o = po.Oracle(path)
    assert ix.info.total_length == o.total_length == len(text) + 1
    plen_r, flat_r = tg.p_rand(20, 50000, 7)
    plen_h, flat_h = tg.p_hit(20, 20, 50000, 8, text)
    plen = np.concatenate([plen_r, plen_h])
    flat = np.concatenate([flat_r, flat_h])
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    of, ol = o.count_flat(plen, flat, starts, threads=8)
    assert np.array_equal(first, of) and np.array_equal(last, ol)
    assert ((last - first + 1)[50000:] >= 1).all()          # sampled substrings always occur
    noccs, offs = ix.locate_flat(plen, flat, starts, 50)
    on, oo = o.locate_flat(plen, flat, starts, 50, threads=8)
    assert np.array_equal(noccs, on) and np.array_equal(offs, oo)
    # located offsets really are occurrences of the pattern in the text
    pos = np.concatenate([[0], np.cumsum(noccs)])
    for i in list(range(0, 200)) + list(range(50000, 50200)):
        p = (flat[starts[i]:starts[i] + plen[i]] - 5).astype(np.uint8)
        for off in offs[pos[i]:pos[i + 1]]:
            assert np.array_equal(text[off:off + len(p)], p)


@pytest.mark.parametrize("mode", [3, 4, 1])
def test_batch_above_a_million_patterns(tmp_path, gpu_ok, mode):
    """Batches above 2^20 patterns are suffix-sorted on the leading symbols only (a partial-bit radix sort,
    query_sort.hip) and, in mode 3, searched from the sorted keys: 1.5 M mixed-length patterns, some longer than a
    key holds, some with characters outside the text's alphabet, against the oracle."""
    text = tg.t_acgt(1 << 21, 31)
    path = _random_index(tmp_path, text, None, "acgt2m")
    ix = _open(path, mode)
    o = po.Oracle(path)
    rng = np.random.Generator(np.random.PCG64(77))
    n = 1_500_000
    plen = rng.integers(0, 30, n).astype(np.int32)           # 0..29 symbols: a key holds 21
    starts = tg.starts_of(plen)
    flat = (np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(plen.sum()))].astype(np.uint16) + 5)
    odd = rng.integers(0, len(flat), 20000)                  # sprinkle characters that do not occur in the text
    flat[odd] = rng.integers(0, 261, len(odd)).astype(np.uint16)
    first, last = ix.count_flat(plen, flat, starts)
    of, ol = o.count_flat(plen, flat, starts, threads=16)
    assert np.array_equal(first, of) and np.array_equal(last, ol)
    noccs, offs = ix.locate_flat(plen, flat, starts, 3)
    on, oo = o.locate_flat(plen, flat, starts, 3, threads=16)
    assert np.array_equal(noccs, on) and np.array_equal(offs, oo)
    # the same batch through the host-pointer pipeline in its other forms: pointer array (parallel_count's
    # alpha_t**), counts only (last == NULL), patterns stored in reverse order (starts not monotone)
    L = femto_amd.lib()
    m = 400_000
    addr = flat.ctypes.data + starts[:m] * 2
    parr = (C.c_void_p * m)(*[int(x) for x in addr])
    pl = np.ascontiguousarray(plen[:m])
    f2 = np.zeros(m, dtype=np.int64)
    l2 = np.zeros(m, dtype=np.int64)
    assert L.femto_amd_parallel_count(ix.handle, m, pl.ctypes.data, parr, f2.ctypes.data, l2.ctypes.data) == 0
    assert np.array_equal(f2, of[:m]) and np.array_equal(l2, ol[:m])
    assert L.femto_amd_parallel_count(ix.handle, m, pl.ctypes.data, parr, f2.ctypes.data, None) == 0
    assert np.array_equal(f2, (ol - of + 1)[:m])
    noccs_p = np.zeros(m, dtype=np.int32)
    offs_p = (C.POINTER(C.c_int64) * m)()
    assert L.femto_amd_parallel_locate(ix.handle, m, pl.ctypes.data, parr, 3, noccs_p.ctypes.data, offs_p) == 0
    assert np.array_equal(noccs_p, on[:m])
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    pos = np.concatenate([[0], np.cumsum(on[:m])])
    for i in range(m):
        if noccs_p[i]:
            if i % 97 == 0:
                assert [offs_p[i][j] for j in range(noccs_p[i])] == list(oo[pos[i]:pos[i + 1]])
            libc.free(offs_p[i])
        else:
            assert not offs_p[i]
    order = np.arange(n)[::-1]
    fr, lr = ix.count_flat(np.ascontiguousarray(plen[order]), flat, np.ascontiguousarray(starts[order]))
    assert np.array_equal(fr, of[order]) and np.array_equal(lr, ol[order])
    bad = plen.copy()
    bad[n // 2] = -1
    with pytest.raises(femto_amd.FemtoAmdError) as ei:
        ix.count_flat(bad, flat, starts)
    assert ei.value.code == 3
    first3, last3 = ix.count_flat(plen, flat, starts)          # the handle stays usable
    assert np.array_equal(first3, of) and np.array_equal(last3, ol)


@pytest.mark.parametrize("sigma", [4, 60])
@pytest.mark.parametrize("mode", [3, 4, 1])
def test_long_patterns_text_tail(tmp_path, gpu_ok, mode, sigma):
    """Patterns much longer than a sort key, on a three-document text: once their range is one row the packed modes
    compare the tail with the text (text_kernels.hip.hpp).  Exact reads, reads with one substituted / inserted symbol
    (the search must die with the reference's values at the reference's step), reads running over a document boundary
    (they contain SEOF), reads from the first positions of the text, random long strings -- all against the oracle."""
    rng = np.random.Generator(np.random.PCG64(900 + sigma))
    alphabet = (np.frombuffer(b"ACGT", dtype=np.uint8) if sigma == 4 else rng.choice(np.arange(32, 127), sigma, replace=False).astype(np.uint8))
    n = 1_500_000
    text = alphabet[rng.integers(0, len(alphabet), n)]
    text[700_000:700_300] = text[100_000:100_300]                    # a repeat: long patterns with two occurrences
    cuts = [400_000, 1_000_000]
    docs = np.split(text, cuts)
    path = str(tmp_path / "longp")
    femto_amd.build_index(path, docs, params="block_size=262144,bucket_size=65536,mark_period=20", infos=["a", "b", "c"], device=0)
    prepared = np.concatenate([np.concatenate([d.astype(np.uint16) + 5, [2]]) for d in docs])
    if mode == 3 and sigma != 4:
        pytest.skip("packed lines need <= 8 characters")
    ix = _open(path, mode)
    o = po.Oracle(path)
    pats = []
    N = len(prepared)
    for _ in range(3000):
        ln = int(rng.integers(17, 160))
        s0 = int(rng.integers(0, N - ln))
        p_ = prepared[s0:s0 + ln].copy()                              # may run over a document boundary (contains SEOF)
        kind = rng.integers(0, 5)
        if kind == 1:
            p_[int(rng.integers(0, ln))] = alphabet[int(rng.integers(0, len(alphabet)))] + 5    # substitution anywhere
        elif kind == 2:
            p_ = np.insert(p_, int(rng.integers(0, ln)), alphabet[int(rng.integers(0, len(alphabet)))] + 5)
        elif kind == 3:
            p_ = prepared[:ln].copy() if rng.random() < 0.5 else prepared[int(rng.integers(0, 40)):][:ln].copy()   # text start
        elif kind == 4:
            p_[int(rng.integers(0, ln))] = int(rng.choice([2, 3, 200, 260]))                       # SEOF / absent characters
        pats.append(p_.astype(np.uint16))
    pats.append(prepared[100_000:100_300].astype(np.uint16))          # the repeat: two rows all the way
    pats += [pats[i % 3000] for i in range(3001)]                     # 6 002 patterns: above the sort threshold
    plen, flat, starts = femto_amd.flatten(pats)
    first, last = ix.count_flat(plen, flat, starts)
    of, ol = o.count_flat(plen, flat, starts, threads=16)
    assert np.array_equal(first, of) and np.array_equal(last, ol)
    assert (ol[:3000] >= of[:3000]).sum() > 500 and (ol[:3000] < of[:3000]).sum() > 500      # both outcomes well represented
    noccs, offs = ix.locate_flat(plen, flat, starts, 4)
    on, oo = o.locate_flat(plen, flat, starts, 4, threads=16)
    assert np.array_equal(noccs, on) and np.array_equal(offs, oo)
    assert on[3000] == 2
    # the row-free form of the device chain (parallel_locate's own results): the text tail locates by its compare, a mismatch ends
    # the pattern at once -- unless the symbol is one the step itself must look at (SEOF, absent characters) or the text starts
    assert_row_free_equals(ix, plen, flat, starts, 4, on, oo, (mode, sigma))
    if mode == (3 if sigma == 4 else 4):      # ... and on the sampled arrays (count_tail_kernel hands plan_rows_kernel the position)
        ix.close()
        ix = femto_amd.Index(path, device=0, options=dict(dense_arrays=0))
        assert not ix.pack_info()["sa_full"] and ix.rank_mode == mode
        assert_row_free_equals(ix, plen, flat, starts, 4, on, oo, (mode, sigma, "sampled"))
        f2, l2 = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f2, of) and np.array_equal(l2, ol)


@pytest.mark.parametrize("mode", MODES)
def test_gpu_built_english_like_vs_oracle(tmp_path, gpu_ok, mode):
    """sigma ~ 96 text (RLE-heavy wavelet nodes, deep Huffman codes), mixed-length patterns 8..64
    (BASELINE config 3 shape, scaled)."""
    text = tg.t_eng(3 << 20, 99)
    path = _random_index(tmp_path, text, "block_size=2097152,bucket_size=262144,mark_period=20", "eng3m")
    ix = femto_amd.Index(path, device=0)
    assert ix.rank_mode == 4 and not ix.pack_info()["available"]    # byte alphabet: the two-level lines are the default
    _set_mode(ix, mode)
    o = po.Oracle(path)
    plen, flat = tg.p_hit(8, 64, 40000, 5, text)
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    of, ol = o.count_flat(plen, flat, starts, threads=8)
    assert np.array_equal(first, of) and np.array_equal(last, ol)
    noccs, offs = ix.locate_flat(plen, flat, starts, 20)
    on, oo = o.locate_flat(plen, flat, starts, 20, threads=8)
    assert np.array_equal(noccs, on) and np.array_equal(offs, oo)


def test_context_table_matches_steps(tmp_path, gpu_ok, monkeypatch):
    """Byte alphabets: the hashed H-gram table (ctx_kernels.hip.hpp) answers the first H steps; the same handle opened
    with FEMTO_AMD_CTX=0 steps through them.  Identical (first, last) -- including those of EMPTY ranges, which are the
    values of the step that emptied them -- for sampled substrings, random strings, patterns shorter than H, patterns
    crossing a document end and patterns holding a character the text lacks; both against the oracle."""
    text = tg.t_eng(2 << 20, 7)
    docs = [text[:700000], text[700000:]]
    path = str(tmp_path / "ctx2doc")
    femto_amd.build_index(path, docs, params="block_size=1048576,bucket_size=131072,mark_period=16", infos=["a", "b"], device=0)
    ix = femto_amd.Index(path, device=0)
    pi = ix.pack_info()
    assert ix.rank_mode == 4 and pi["sa_full"] and pi["context_table"] and 5 <= pi["context_syms"] <= 12, pi
    assert pi["context_syms"] < pi["context2_syms"] <= 16, pi      # ... and the wide table behind it
    H = pi["context2_syms"]
    rng = np.random.Generator(np.random.PCG64(77))
    plen, flat = tg.p_hit(1, 40, 30000, 9, text)
    pats = [flat[s:s + l] for s, l in zip(tg.starts_of(plen), plen)]
    alphabet = np.unique(text)
    for _ in range(8000):                              # random strings over the text's alphabet: most die inside the H steps
        pats.append(tg.to_alpha(alphabet[rng.integers(0, len(alphabet), int(rng.integers(1, 20)))]))
    for _ in range(2000):                              # a sampled substring with one symbol replaced
        l = int(rng.integers(H, 30))
        s0 = int(rng.integers(0, len(text) - l))
        q = text[s0:s0 + l].copy()
        q[int(rng.integers(0, l))] = alphabet[int(rng.integers(0, len(alphabet)))]
        pats.append(tg.to_alpha(q))
    missing = [c for c in range(256) if c not in set(alphabet.tolist())][:3]
    for c in missing:                                  # a character the text lacks, inside and outside the last H symbols
        for pos in (0, 3, 12):
            q = text[5000:5020].copy()
            q[pos] = c
            pats.append(tg.to_alpha(q))
    for cut in (699990, 699995):                       # across the document end: SEOF (alpha code 2) inside the pattern
        q = np.concatenate([tg.to_alpha(text[cut:700000]), np.array([2], dtype=np.uint16), tg.to_alpha(text[700000:700000 + 12])])
        pats.append(q)
    plen, flat, starts = femto_amd.flatten(pats)
    first, last = ix.count_flat(plen, flat, starts)
    noccs, offs = ix.locate_flat(plen, flat, starts, 10)
    ix.close()
    monkeypatch.setenv("FEMTO_AMD_CTX", "0")
    monkeypatch.setenv("FEMTO_AMD_TAIL_ROWS", "4")     # ... and the text tail taken by ranges of up to four rows
    monkeypatch.setenv("FEMTO_AMD_TAIL_ROW_COST", "1")
    ix0 = femto_amd.Index(path, device=0)
    assert not ix0.pack_info()["context_table"]
    f0, l0 = ix0.count_flat(plen, flat, starts)
    n0, o0 = ix0.locate_flat(plen, flat, starts, 10)
    ix0.close()
    assert np.array_equal(first, f0) and np.array_equal(last, l0)
    assert np.array_equal(noccs, n0) and np.array_equal(offs, o0)
    o = po.Oracle(path)
    of, ol = o.count_flat(plen, flat, starts, threads=8)
    assert np.array_equal(first, of) and np.array_equal(last, ol)
    # (a sampled substring misses only when it straddles the cut between the two documents)
    assert (last[:30000] >= first[:30000]).sum() > 29900 and (last[30000:38000] < first[30000:38000]).sum() > 4000


def test_full_text_lf_walk_recovers_every_offset(tmp_path, gpu_ok):
    """Size-independent property: locating the range of the EMPTY pattern (all rows) returns a
    permutation of 0..n-1, i.e. the whole suffix array, and L[row] == text[SA[row]-1]."""
    text = tg.t_acgt(300000, 5)
    path = _random_index(tmp_path, text, "block_size=131072,bucket_size=16384,mark_period=32", "perm")
    ix = femto_amd.Index(path, device=0)
    n = ix.info.total_length
    noccs, offs = ix.locate([np.zeros(0, dtype=np.uint16)], n)
    assert noccs[0] == n
    assert np.array_equal(np.sort(offs), np.arange(n))
    ch, _, _ = ix.block_requests(np.arange(n, dtype=np.int64))
    prepared = np.concatenate([text.astype(np.uint16) + 5, [2]])
    assert np.array_equal(ch, prepared[offs - 1])           # SA[row]==0 wraps to the final SEOF


@pytest.mark.parametrize("seed", range(int(os.environ.get("FEMTO_AMD_SWEEP_SEEDS", "40"))))
def test_random_indexes_vs_oracle(tmp_path, gpu_ok, seed):
    """Randomised parity sweep: random alphabets / run structure / document splits / index parameters,
    index built on the GPU (suffix sorter + writer), then count, locate (random clamps), leaf requests and
    LF^-1 steps compared with the oracle, in every kernel mode."""
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    n = int(rng.integers(2000, 60000))
    sigma = int(rng.choice([1, 2, 3, 4, 8, 20, 64, 200, 256]))
    alphabet = rng.choice(256, sigma, replace=False).astype(np.uint8)
    if rng.random() < 0.5:      # skewed, run-heavy text (RLE segments, single-character buckets)
        runs = rng.geometric(1.0 / float(rng.choice([2, 20, 400])), n)
        syms = alphabet[rng.integers(0, sigma, n)]
        text = np.repeat(syms, runs)[:n]
    else:
        text = alphabet[rng.integers(0, sigma, n)]
    ndocs = int(rng.integers(1, 5))
    cuts = sorted(rng.choice(np.arange(1, len(text)), ndocs - 1, replace=False)) if ndocs > 1 else []
    docs = np.split(text, cuts)
    b_size = int(rng.choice([64, 100, 1000, 4096, 1 << 20]))
    block = b_size * int(rng.choice([1, 2, 5]))
    mark = int(rng.integers(1, 40))
    params = f"block_size={block},bucket_size={b_size},chunk_size={b_size},mark_period={mark}"
    path = str(tmp_path / f"rnd{seed}")
    femto_amd.build_index(path, docs, params=params, infos=[f"d{i}" for i in range(len(docs))], device=0)
    # the GPU suffix sorter against a CPU suffix array of the same prepared text: identical index files
    from sa_util import suffix_array
    import filecmp
    prepared = np.concatenate([np.concatenate([d.astype(np.uint16) + 5, [2]]) for d in docs])
    ref_path = str(tmp_path / f"rnd{seed}_cpu_sa")
    femto_amd.build_index_from_sa(ref_path, docs, suffix_array(prepared), params=params, infos=[f"d{i}" for i in range(len(docs))])
    for f in sorted(os.listdir(ref_path)):
        if f != "_femto_index":
            assert filecmp.cmp(os.path.join(path, f), os.path.join(ref_path, f), shallow=False), (seed, f, params)
    o = po.Oracle(path)
    ix = femto_amd.Index(path, device=0)
    distinct = len(np.unique(text)) + 1          # + SEOF
    assert ix.rank_mode == (3 if distinct <= 8 else 4 if distinct <= 256 else 1)
    ix.close()
    ix = _open(path)
    nrows = ix.info.total_length
    assert nrows == o.total_length == len(text) + len(docs)
    pats = []
    for _ in range(300):
        l = int(rng.integers(0, 30))
        if rng.random() < 0.6 and len(text) > l:
            s0 = int(rng.integers(0, len(text) - l + 1))
            pats.append(tg.to_alpha(text[s0:s0 + l]))
        else:
            pats.append(tg.to_alpha(rng.integers(0, 256, l).astype(np.uint8)))
    plen, flat, starts = femto_amd.flatten(pats)
    of, ol = o.count_flat(plen, flat, starts)
    mo = int(rng.integers(1, 50))
    on, oo = o.locate_flat(plen, flat, starts, mo)
    rows = rng.integers(0, nrows, 500).astype(np.int64)
    want_fw = [o.forward_step(int(r)) for r in rows]
    want_bw = [o.block_request(int(r), 7) for r in rows]
    assert ix.pack_info()["available"] == (distinct <= 8) and ix.pack_info()["available2"] == (distinct <= 256)
    for mode in MODES:
        if (mode == 3 and distinct > 8) or (mode == 4 and distinct > 256):
            continue
        ix.set_rank_mode(mode)
        f, l_ = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f, of) and np.array_equal(l_, ol), (seed, mode, params)
        nn, offs = ix.locate_flat(plen, flat, starts, mo)
        assert np.array_equal(nn, on) and np.array_equal(offs, oo), (seed, mode, params)
        ch, occ, off = ix.block_requests(rows)
        assert [(int(a), int(b), int(c)) for a, b, c in zip(ch, occ, off)] == want_bw, (seed, mode)
    ch, nr, off = ix.forward_steps(rows)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(ch, nr, off)] == want_fw, seed


@pytest.mark.parametrize("large", [False, True])
@pytest.mark.parametrize("ndocs", [1, 3])
def test_gpu_sorter_full_byte_alphabet(tmp_path, gpu_ok, ndocs, large, monkeypatch):
    """Texts that use (almost) every byte value: 256-257 symbols with SEOF, which do not fit 8-bit ranks (an
    earlier 8-bit rank table wrapped the last symbol onto the end marker).  Both sorter paths, checked through the
    byte identity of the index with the one built from a CPU suffix array, and through locate-all == that array."""
    import filecmp
    from sa_util import suffix_array
    rng = np.random.Generator(np.random.PCG64(41 + ndocs))
    n = 150_000
    text = rng.integers(0, 256, n).astype(np.uint8)
    assert len(np.unique(text)) == 256
    cuts = sorted(rng.choice(np.arange(1, n), ndocs - 1, replace=False)) if ndocs > 1 else []
    docs = np.split(text, cuts)
    params = "block_size=65536,bucket_size=4096,chunk_size=4096,mark_period=8"
    if large:
        monkeypatch.setenv("FEMTO_AMD_LARGE_SORT_CAP", "40000")
    a, b = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    femto_amd.build_index(a, docs, params=params, infos=[f"d{i}" for i in range(len(docs))], device=0)
    prepared = np.concatenate([np.concatenate([d.astype(np.uint16) + 5, [2]]) for d in docs])
    sa = suffix_array(prepared)
    femto_amd.build_index_from_sa(b, docs, sa, params=params, infos=[f"d{i}" for i in range(len(docs))])
    for f in sorted(os.listdir(b)):
        if f != "_femto_index":
            assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f
    if ndocs == 1:
        ix = femto_amd.Index(a, device=0)
        assert ix.rank_mode == 1                     # 257 symbols: the wavelet path
        _, offs = ix.locate([np.zeros(0, dtype=np.uint16)], len(sa))
        assert np.array_equal(offs, sa)


@pytest.mark.parametrize("kind", ["acgt", "eng", "runs"])
def test_large_text_suffix_sorter_path(tmp_path, gpu_ok, kind, monkeypatch):
    """The 64-bit, partitioned suffix sorter used for texts of 2^32 symbols and more, forced onto small
    inputs (FEMTO_AMD_LARGE_SORT_CAP = part capacity) and checked through the byte-identity of the index it
    yields with the index built by the 32-bit sorter, and against numpy's suffix array."""
    from sa_util import suffix_array
    if kind == "acgt":
        text = tg.t_acgt(200000, 77)
        cap = 60000
    elif kind == "eng":
        text = tg.t_eng(150000, 78)
        cap = 150002
    else:
        rng = np.random.Generator(np.random.PCG64(5))
        text = np.repeat(rng.choice(np.frombuffer(b"ab", dtype=np.uint8), 3000), rng.integers(1, 60, 3000)).astype(np.uint8)
        cap = len(text) + 2
    params = "block_size=65536,bucket_size=8192,mark_period=20"
    a, b = str(tmp_path / "small"), str(tmp_path / "large")
    femto_amd.build_index(a, [text], params=params, infos=["x"], device=0)
    monkeypatch.setenv("FEMTO_AMD_LARGE_SORT_CAP", str(cap))
    femto_amd.build_index(b, [text], params=params, infos=["x"], device=0)
    monkeypatch.delenv("FEMTO_AMD_LARGE_SORT_CAP")
    import filecmp
    files = sorted(f for f in os.listdir(a) if f != "_femto_index")
    assert files == sorted(f for f in os.listdir(b) if f != "_femto_index")
    for f in files:
        assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f
    # and the suffix array itself, read back through locate of the empty pattern
    ix = femto_amd.Index(b, device=0)
    n = ix.info.total_length
    _, offs = ix.locate([np.zeros(0, dtype=np.uint16)], n)
    sa = suffix_array(np.concatenate([text.astype(np.uint16) + 5, [2]]))
    assert np.array_equal(offs, sa)


def test_config0_16mib_vs_genuine_reference(tmp_path, gpu_ok):
    """BASELINE configs[0], the reference's own CPU-runnable case: 16 MiB random-ACGT text, default index parameters,
    100 k 20-mers (half sampled from the text, half random -- the mix BASELINE.md measured).  The index is built by this
    repo's builder; count and locate on the GPU must equal the GENUINE reference's parallel_count / parallel_locate on
    the same files (oracle/_ref/ref_tool, prebuilt where /root/reference exists) and the oracle port."""
    n = 1 << 24
    text = tg.t_acgt(n, 160)
    path = str(tmp_path / "acgt16m")
    femto_amd.build_index(path, [text], params=None, infos=["cfg0"], device=0)
    ix = femto_amd.Index(path, device=0)
    assert ix.info.total_length == n + 1 and ix.info.number_of_blocks == 1 and ix.info.total_buckets == 17
    hp, hf = tg.p_hit(20, 20, 50_000, 5, text)
    rp, rf = tg.p_rand(20, 50_000, 6)
    plen = np.concatenate([hp, rp])
    flat = np.concatenate([hf, rf])
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    assert (last[:50_000] >= first[:50_000]).all()
    noccs, offs = ix.locate_flat(plen, flat, starts, 100)
    o = po.Oracle(path)
    of, ol = o.count_flat(plen, flat, starts, threads=16)
    on, oo = o.locate_flat(plen, flat, starts, 100, threads=16)
    assert np.array_equal(first, of) and np.array_equal(last, ol) and np.array_equal(noccs, on) and np.array_equal(offs, oo)
    if po.have_ref():
        pf = str(tmp_path / "p.fpat")
        po.write_fpat_flat(pf, plen, flat)
        po.ref_tool("count", path, pf, str(tmp_path / "c.bin"), capture=False)
        r = np.fromfile(str(tmp_path / "c.bin"), dtype=np.int64)
        assert np.array_equal(r[:len(plen)], first) and np.array_equal(r[len(plen):], last)
        po.ref_tool("locate", path, pf, 100, str(tmp_path / "l.bin"), capture=False)
        raw = np.fromfile(str(tmp_path / "l.bin"), dtype=np.uint8)
        assert np.array_equal(raw[:4 * len(plen)].view(np.int32), noccs)
        assert np.array_equal(raw[4 * len(plen):].view(np.int64), offs)
    ix.close()

"""`-m gpu` parity tests on the committed fixtures: the HIP path (through the C ABI, femto_amd/libfemto_amd.so) against the
committed golden vectors of the genuine reference -- every row, every pattern, every kernel family, every option set.
Bit-exact (integer work).  Siblings: test_gpu_built.py (indexes built here vs the oracle), test_gpu_fullsize.py (BASELINE's
full sizes), test_gpu_multi.py (several handles / processes on this box's GPU), test_gpu_multidevice.py (real peers),
test_gpu_cli.py (the femto_search counterpart)."""
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


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_leaf_requests_match_reference(fixtures, gpu_ok, name, mode):
    """block_request CHAR|OCCS|LOCATION for every row (index_test.c:60-476 checks the same leaves)."""
    fx = fixtures(name)
    g = fx.gold
    ix = _open(fx.index, mode)
    n = ix.info.total_length
    rows = np.arange(n, dtype=np.int64)
    ch, occ, off = ix.block_requests(rows)
    assert np.array_equal(ch, g["L"])
    assert np.array_equal(occ, g["occ"])
    assert np.array_equal(off, g["off"])
    for key in g.files:
        if key.startswith("occs_ch"):
            c = int(key[7:])
            _, occ_c, _ = ix.block_requests(rows, np.full(n, c, dtype=np.uint16))
            assert np.array_equal(occ_c, g[key]), c
    ix.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_count_locate_match_reference_goldens(fixtures, gpu_ok, name, mode):
    fx = fixtures(name)
    ix = _open(fx.index, mode)
    plen, flat, starts = fx.patterns
    first, last = ix.count_flat(plen, flat, starts)
    assert np.array_equal(first, fx.gold["count_first"])
    assert np.array_equal(last, fx.gold["count_last"])
    for mo, noccs, offs in fx.locate_cases():
        n, got = ix.locate_flat(plen, flat, starts, mo)                # one-pass form (femto_amd_locate_flat_alloc)
        assert np.array_equal(n, noccs), mo
        assert np.array_equal(got, offs), mo
        n, got = ix.locate_flat_two_call(plen, flat, starts, mo)       # sizing call + fill call (femto_amd_locate_flat)
        assert np.array_equal(n, noccs), mo
        assert np.array_equal(got, offs), mo
    ix.close()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_forward_steps_match_reference(fixtures, gpu_ok, name):
    """do_forward_query (LF^-1 via bseq_select / wtree_select) for every row, against the leaf answers
    captured from the reference; and LF(LF^-1(row)) == row."""
    fx = fixtures(name)
    g = fx.gold
    ix = femto_amd.Index(fx.index, device=0)
    n = ix.info.total_length
    rows = np.arange(n, dtype=np.int64)
    ch, nr, off = ix.forward_steps(rows)
    assert np.array_equal(ch, g["fwd_ch"])
    assert np.array_equal(nr, g["fwd_row"])
    assert np.array_equal(off, g["fwd_off"])
    valid = nr >= 0
    lch, locc, _ = ix.block_requests(nr[valid])
    assert np.array_equal(lch, ch[valid])     # L[LF^-1(r)] == F[r]


def test_flattened_index(fixtures, gpu_ok):
    fx = fixtures("acgt48k")
    a = femto_amd.Index(fx.index, device=0)
    b = femto_amd.Index(fx.flat, device=0)
    plen, flat, starts = fx.patterns
    for x, y in zip(a.count_flat(plen, flat, starts), b.count_flat(plen, flat, starts)):
        assert np.array_equal(x, y)
    for x, y in zip(a.locate_flat(plen, flat, starts, 9), b.locate_flat(plen, flat, starts, 9)):
        assert np.array_equal(x, y)


def test_pointer_array_forms(fixtures, gpu_ok):
    """femto_amd_parallel_count / femto_amd_parallel_locate: the reference's own calling convention
    (alpha_t** patterns, callee-malloc'd offsets[i], femto.c:275-400)."""
    fx = fixtures("eng2doc")
    ix = femto_amd.Index(fx.index, device=0)
    plen, flat, starts = fx.patterns
    n = len(plen)
    L = femto_amd.lib()
    pats = [np.ascontiguousarray(flat[starts[i]:starts[i] + plen[i]]) for i in range(n)]
    parr = (C.c_void_p * n)(*[p.ctypes.data if len(p) else None for p in pats])
    pl = plen.astype(np.int32)
    first = np.zeros(n, dtype=np.int64)
    last = np.zeros(n, dtype=np.int64)
    assert L.femto_amd_parallel_count(ix.handle, n, pl.ctypes.data, parr, first.ctypes.data, last.ctypes.data) == 0
    assert np.array_equal(first, fx.gold["count_first"]) and np.array_equal(last, fx.gold["count_last"])
    cnt = np.zeros(n, dtype=np.int64)   # last == NULL -> counts (femto.c:313-318)
    assert L.femto_amd_parallel_count(ix.handle, n, pl.ctypes.data, parr, cnt.ctypes.data, None) == 0
    assert np.array_equal(cnt, last - first + 1)
    noccs = np.zeros(n, dtype=np.int32)
    offs = (C.POINTER(C.c_int64) * n)()
    assert L.femto_amd_parallel_locate(ix.handle, n, pl.ctypes.data, parr, 7, noccs.ctypes.data, offs) == 0
    assert np.array_equal(noccs, fx.gold["loc7_noccs"])
    got = []
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for i in range(n):
        if noccs[i]:
            got.extend(offs[i][j] for j in range(noccs[i]))
            libc.free(offs[i])
        else:
            assert not offs[i]
    assert np.array_equal(np.array(got, dtype=np.int64), fx.gold["loc7_offs"])


def test_concurrent_callers_on_one_handle(tmp_path, gpu_ok):
    """The reference accepts blocking calls from many threads at once (src/main/server.c:3732-3793): every call here
    leases its own scratch (buffers, flags, stream), so four host threads issuing count / locate batches on ONE handle
    overlap on the GPU and must each get exactly the answers of a serial run -- host-pointer and device-pointer forms."""
    import threading
    import torch
    from femto_amd import textgen as tg
    text = tg.t_acgt(1 << 20, 77)
    path = str(tmp_path / "ix")
    femto_amd.build_index(path, [text], infos=["t"])
    ix = femto_amd.Index(path, device=0)
    batches = []
    for t in range(4):
        plen, flat = tg.p_hit(6, 30, 300_000 + 1000 * t, 1000 + t, text)   # above the pipelined-staging threshold too
        batches.append((plen, flat, tg.starts_of(plen)))
    serial = [(ix.count_flat(*b), ix.locate_flat(*b, 5)) for b in batches]
    results, errors = [None] * 4, []

    def work(t):
        try:
            out = []
            for _ in range(3):
                out.append((ix.count_flat(*batches[t]), ix.locate_flat(*batches[t], 5)))
            results[t] = out
        except Exception as ex:      # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        (sf, sl), (sn, so) = serial[t]
        for (f, l), (n, o) in results[t]:
            assert np.array_equal(f, sf) and np.array_equal(l, sl) and np.array_equal(n, sn) and np.array_equal(o, so)
    # enqueue-only calls on four different streams at once (the scratch of one must not be reused by another in flight)
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for t, (plen, flat, starts) in enumerate(batches):
        n = len(plen)
        d = {"plen": torch.from_numpy(plen).to(dev), "flat": torch.from_numpy(flat.view(np.int16)).to(dev),
             "starts": torch.from_numpy(starts).to(dev), "res": torch.empty((2, n), dtype=torch.int64, device=dev),
             "noccs": torch.empty(n, dtype=torch.int32, device=dev), "ost": torch.empty(n + 1, dtype=torch.int64, device=dev),
             "offs": torch.empty(len(serial[t][1][1]) + 16, dtype=torch.int64, device=dev), "tot": torch.zeros(2, dtype=torch.int64, device=dev)}
        outs.append(d)
    torch.cuda.synchronize()
    for rep in range(3):
        for t, d in enumerate(outs):
            n = len(batches[t][0])
            ix.locate_device(n, d["plen"].data_ptr(), d["flat"].data_ptr(), d["starts"].data_ptr(), 5, d["res"][0].data_ptr(),
                             d["res"][1].data_ptr(), d["noccs"].data_ptr(), d["ost"].data_ptr(), d["offs"].data_ptr(), d["offs"].numel(),
                             d["tot"].data_ptr(), streams[t].cuda_stream)
    torch.cuda.synchronize()
    for t, d in enumerate(outs):
        (sf, sl), (sn, so) = serial[t]
        tot = d["tot"].cpu().numpy()
        assert tot[1] == 0 and tot[0] == len(so)
        assert np.array_equal(d["res"][0].cpu().numpy(), sf) and np.array_equal(d["res"][1].cpu().numpy(), sl)
        assert np.array_equal(d["noccs"].cpu().numpy(), sn) and np.array_equal(d["offs"][:len(so)].cpu().numpy(), so)
    ix.close()


@pytest.mark.parametrize("mode", MODES)
def test_locate_range_matches_reference(fixtures, gpu_ok, mode):
    """parallel_locate_range (femto.c:481): every row's offset, against the per-row LOCATION answers captured from the
    reference walked back to a mark -- here simply against locate of the empty pattern, which the goldens pin."""
    fx = fixtures("eng2doc")
    ix = _open(fx.index, mode)
    n = ix.info.total_length
    _, every = ix.locate([np.zeros(0, dtype=np.uint16)], n)
    assert np.array_equal(ix.locate_range(0, n - 1), every)
    assert np.array_equal(ix.locate_range(123, 4567), every[123:4568])
    marked = fx.gold["off"] >= 0                      # rows the reference marks answer with their own offset
    assert np.array_equal(every[marked], fx.gold["off"][marked])
    for bad in [(-1, 5), (7, 3), (0, n)]:
        with pytest.raises(femto_amd.FemtoAmdError) as ei:
            ix.locate_range(*bad)
        assert ei.value.code == 3
    ix.close()


def test_invalid_pattern_character_is_param_error(fixtures, gpu_ok):
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=0)
    with pytest.raises(femto_amd.FemtoAmdError) as ei:
        ix.count([np.array([70, 300], dtype=np.uint16)])
    assert ei.value.code == 3
    # the handle stays usable
    f, l = ix.count([np.array([70], dtype=np.uint16)])
    assert l[0] >= f[0]
    # enqueue-only calls report nothing to the host: the bad pattern has the empty range there, and a host-pointer call
    # that follows on the same scratch does not inherit the flag its kernels raised
    import torch
    for mode in MODES:
        if mode == 4:
            continue
        ix.set_rank_mode(mode)
        plen = torch.tensor([2, 1], dtype=torch.int32, device="cuda:0")
        flat = torch.tensor([70, 300, 70], dtype=torch.int16, device="cuda:0")
        starts = torch.tensor([0, 2], dtype=torch.int64, device="cuda:0")
        res = torch.full((2, 2), 7, dtype=torch.int64, device="cuda:0")
        ix.count_device(2, plen.data_ptr(), flat.data_ptr(), starts.data_ptr(), res[0].data_ptr(), res[1].data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r = res.cpu().numpy()
        assert r[0][0] > r[1][0] and (r[0][1], r[1][1]) == (f[0], l[0]), (mode, r)
        f2, l2 = ix.count([np.array([70], dtype=np.uint16)])
        assert (f2[0], l2[0]) == (f[0], l[0]), mode


def test_max_occs_zero_mirrors_reference(fixtures, gpu_ok):
    """parallel_locate with max_occs_each == 0: the reference returns a single match whole but fails with
    ERR_PARAM as soon as one pattern has more than one match (observed with oracle/_ref: "invalid
    parameters: Error during query processing", femto.c:204)."""
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=0)
    text = fx.docs[0]
    single = tg.to_alpha(text[1000:1024])          # a 24-mer of a 48 KiB random text: exactly one match
    f, l = ix.count([single])
    assert l[0] - f[0] == 0
    noccs, offs = ix.locate([single], 0)
    assert noccs[0] == 1 and offs[0] == 1000
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        ix.locate([tg.to_alpha(text[:2])], 0)         # a 2-mer: many matches
    assert e.value.code == 3
    with pytest.raises(femto_amd.FemtoAmdError):
        ix.locate([single], -1)


def test_pack_counts_device(fixtures, gpu_ok):
    """femto_amd_pack_counts_device: one byte per match count, (pattern, count) pairs for 255 and more, overflow reported"""
    import torch
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=0)
    rng = np.random.Generator(np.random.PCG64(5))
    n = 100_000
    first = rng.integers(0, 1 << 40, n)
    cnt = rng.integers(0, 300, n)
    cnt[rng.integers(0, n, 50)] = rng.integers(1 << 20, 1 << 39, 50)
    last = first + cnt - 1
    last[cnt == 0] = first[cnt == 0] - rng.integers(1, 9, int((cnt == 0).sum()))      # first > last by any amount: no match
    d_f, d_l = torch.from_numpy(first).cuda(), torch.from_numpy(last).cuda()
    c8 = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
    nbig = int((cnt >= 255).sum())
    for cap in (nbig + 10, nbig // 2):
        big = torch.zeros(2 * max(cap, 1), dtype=torch.int64, device="cuda:0")
        bn = torch.full((1,), -1, dtype=torch.int64, device="cuda:0")
        ix.pack_counts_device(n, d_f.data_ptr(), d_l.data_ptr(), c8.data_ptr(), big.data_ptr(), cap, bn.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(bn.item()) == nbig                                   # counted even beyond the capacity
        assert np.array_equal(c8.cpu().numpy(), np.minimum(cnt, 255).astype(np.uint8))
        pairs = big.cpu().numpy().reshape(-1, 2)[:min(cap, nbig)]
        assert len(set(pairs[:, 0].tolist())) == len(pairs) and (cnt[pairs[:, 0]] == pairs[:, 1]).all() and (pairs[:, 1] >= 255).all()
    ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc", "chunks2doc"])
def test_device_chain_walks_inside_the_row_expansion(fixtures, gpu_ok, name):
    """femto_amd_locate_device / _locate_keys_device WITHOUT the resident suffix array: plan_rows_kernel<2> walks every located
    row to its next derived mark inside the row expansion (no rows written, no walk kernel).  Same noccs / out_starts /
    offsets as the reference's goldens for every mark density (femto's own, every 3rd, every 5th), 4- and 8-byte mark offsets,
    a capacity that cuts the output short, and ranges longer than the per-lane limit (the empty pattern with a huge max_occs:
    plan_big_rows_kernel walks those)."""
    import torch
    fx = fixtures(name)
    plen, flat, starts = fx.patterns
    n = len(plen)
    dev = "cuda:0"
    d_plen, d_flat, d_starts = torch.from_numpy(plen).to(dev), torch.from_numpy(flat.view(np.int16)).to(dev), torch.from_numpy(starts).to(dev)
    for kw in (dict(dense_arrays=0), dict(dense_arrays=0, mark_every=0), dict(dense_arrays=0, mark_every=3, marks_32bit=0),
               dict(dense_arrays=0, text=0, rank_units=0), dict(dense_arrays=0, text=0), dict(dense_arrays=0, text=0, mark_every=10),
               dict(hbm_budget_bytes=600_000), dict(hbm_budget_bytes=150_000),
               # the marked rank units ("mark spotting": the search hands plan_rows_kernel a marked row it stood on) against the
               # plain ones, with few table symbols so that most steps run on units, at three mark densities
               dict(dense_arrays=0, text=0, rank_units=2), dict(dense_arrays=0, text=0, rank_units=3, level_table_syms=2),
               dict(text=0, rank_units=3, level_table=0, mark_every=3), dict(dense_arrays=0, rank_units=3, mark_every=0, level_table_syms=1)):
        ix = femto_amd.Index(fx.index, device=0, options=kw)
        assert "hbm_budget_bytes" in kw or not ix.pack_info()["sa_full"]
        if ix.pack_info()["rank_units"]:      # auto: marked exactly where the handle walks (a budget may still pay for a tiny fixture's suffix array)
            want_marked = kw.get("rank_units", 1) == 3 or (kw.get("rank_units", 1) == 1 and not ix.pack_info()["sa_full"])
            # (under a tight budget the plain units are built where the marked ones miss their share: profiles/r05_budget_sweep.txt, 3 x text)
            assert ix.pack_info()["rank_units_marked"] == want_marked or ("hbm_budget_bytes" in kw and not ix.pack_info()["rank_units_marked"]), (kw, ix.pack_info())
        if ix.rank_mode not in (3, 4) or ix.pack_info()["sa_full"]:      # (a budget that still pays for the dense arrays of a tiny fixture)
            ix.close()
            continue
        for mo, g_noccs, g_offs in list(fx.locate_cases()) + [(1 << 20, None, None)]:
            if g_noccs is None:      # a limit nothing reaches: every row of every pattern, the empty pattern's whole index
                g_noccs, g_offs = ix.locate_flat(plen, flat, starts, mo)
            tot = int(g_noccs.astype(np.int64).sum())
            for cap in (tot + 8, max(1, tot // 2)):
                f, l = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
                noccs = torch.zeros(n, dtype=torch.int32, device=dev)
                ostarts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
                offs = torch.full((cap,), -7, dtype=torch.int64, device=dev)
                total = torch.zeros(2, dtype=torch.int64, device=dev)
                for rep in range(2):
                    ix.locate_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), mo, f.data_ptr(), l.data_ptr(), noccs.data_ptr(),
                                     ostarts.data_ptr(), offs.data_ptr(), cap, total.data_ptr())
                    torch.cuda.synchronize()
                    assert total.cpu().tolist() == [tot, 1 if tot > cap else 0], (kw, mo, cap)
                    assert np.array_equal(noccs.cpu().numpy(), g_noccs) and np.array_equal(f.cpu().numpy(), fx.gold["count_first"])
                    assert np.array_equal(l.cpu().numpy(), fx.gold["count_last"])
                    assert np.array_equal(offs.cpu().numpy()[:min(cap, tot)], g_offs[:min(cap, tot)]), (kw, mo, cap, rep)
        ix.close()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_row_free_locate_device(fixtures, gpu_ok, name):
    """femto_amd_locate_device with d_first == d_last == NULL -- parallel_locate's own results, noccs and offsets and no rows
    (src/main/femto.c:331-400): the reference's golden noccs / offsets for every clamp, under option sets that take every path
    the form differs on -- the inline text tail of dense handles (located by the compare, no inverse-suffix-array read; a
    mismatch ends the pattern without the emptying step), count_tail_kernel on the sampled arrays (the position instead of the
    way back to a row; plan_rows_kernel does not walk), handles without the text, mark spotting, and femto's own wavelet tree."""
    fx = fixtures(name)
    plen, flat, starts = fx.patterns
    sets = [dict(), dict(tail_min=2, tail_ones=0), dict(tail_min=2, tail_ones=0, level_table=0, context_table=0), dict(tail_min=2, tail_rows=4, tail_row_cost=0),
            dict(dense_arrays=0), dict(dense_arrays=0, tail_min=2), dict(dense_arrays=0, tail_min=2, mark_every=3, level_table_syms=1),
            dict(dense_arrays=0, text=0), dict(dense_arrays=0, text=0, rank_units=3, level_table_syms=2), dict(hbm_budget_bytes=600_000)]
    for kw in sets:
        ix = femto_amd.Index(fx.index, device=0, options=kw)
        for mode in ([ix.rank_mode] if kw else [ix.rank_mode, 1, 0]):
            ix.set_rank_mode(mode)
            for mo, g_noccs, g_offs in fx.locate_cases():
                assert_row_free_equals(ix, plen, flat, starts, mo, g_noccs, g_offs, (name, kw, mode, mo))
        ix.close()


@pytest.mark.parametrize("env", [dict(FEMTO_AMD_SA32_DENSE="0"), dict(FEMTO_AMD_KTAB_SA1="0"), dict(FEMTO_AMD_SA32_DENSE="0", FEMTO_AMD_KTAB_SA1="0")])
@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_big_index_forms_on_the_fixtures(fixtures, gpu_ok, monkeypatch, name, env):
    """What only indexes above 2^31 / 2^32 rows take by themselves -- 8-byte suffix-array / inverse entries, level-table entries
    without a text position -- forced on the fixtures (the knobs the A/B runs of profiles/tuning_history.md used): goldens for every
    clamp through the host path, the device chain and its row-free form, dense and sampled arrays."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    for kw in (dict(hbm_budget_bytes=femto_amd.BUDGET_ALL), dict(hbm_budget_bytes=femto_amd.BUDGET_ALL, tail_min=2, tail_ones=0), dict(dense_arrays=0, tail_min=2)):
        ix = femto_amd.Index(fx.index, device=0, options=kw)
        pi = ix.pack_info()
        if "FEMTO_AMD_SA32_DENSE" in env and ix.rank_mode in (3, 4):
            assert not pi["sa_32bit"], pi
        f, l = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f, g["count_first"]) and np.array_equal(l, g["count_last"]), (kw, env)
        for mo, g_noccs, g_offs in fx.locate_cases():
            n_, o_ = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(n_, g_noccs) and np.array_equal(o_, g_offs), (kw, env, mo)
            df, dl, dn, dst, do, dtot = device_locate(ix, plen, flat, starts, mo, len(g_offs) + 16)
            assert dtot == len(g_offs) and np.array_equal(df, g["count_first"]) and np.array_equal(dl, g["count_last"]), (kw, env, mo)
            assert np.array_equal(dn, g_noccs) and np.array_equal(do, g_offs), (kw, env, mo)
            assert_row_free_equals(ix, plen, flat, starts, mo, g_noccs, g_offs, (name, kw, env, mo))
        ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc"])
def test_keys_device_path_equals_symbol_path(fixtures, gpu_ok, name):
    """femto_amd_pack_keys_device + femto_amd_locate_keys_device: patterns as 64-bit keys, ranges as int32 pairs -- the same
    (first, last), clamped row counts, out_starts and located offsets as the symbol entry points and the reference's goldens,
    for every pattern a key describes; the others are counted in *d_bad"""
    import torch
    fx = fixtures(name)
    g = fx.gold
    # ... on the default handle and on handles that walk to marks (no suffix array: on small alphabets the marked rank units hand
    # plan_rows_kernel a marked row the key search stood on, as count_direct_kernel does)
    for kw in (None, dict(dense_arrays=0, text=0, level_table_syms=2), dict(text=0, mark_every=3, level_table=0)):
        _keys_device_path(fx, g, femto_amd.Index(fx.index, device=0, options=kw) if kw else femto_amd.Index(fx.index, device=0))


def _keys_device_path(fx, g, ix):
    import torch
    bits, max_syms, table = ix.key_format()
    assert 63 // bits == max_syms and table.max() < (1 << bits)
    plen, flat, starts = fx.patterns
    n = len(plen)
    in_text = table[np.minimum(flat, 260)] != 0
    in_text[flat > 260] = False
    ok = np.array([plen[i] <= max_syms and bool(in_text[starts[i]:starts[i] + plen[i]].all()) for i in range(n)])
    assert ok.sum() >= 20
    dev = "cuda:0"
    d_plen, d_flat, d_starts = torch.from_numpy(plen).to(dev), torch.from_numpy(flat.view(np.int16)).to(dev), torch.from_numpy(starts).to(dev)
    d_keys = torch.zeros(n, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    ix.pack_keys_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), d_keys.data_ptr(), d_bad.data_ptr())
    torch.cuda.synchronize()
    assert int(d_bad.item()) == int((~ok).sum())
    # the keys of the host-side definition: field j from the top = the j-th symbol from the end
    keys = d_keys.cpu().numpy().view(np.uint64)
    for i in np.flatnonzero(ok)[:50]:
        k = 0
        for s in range(plen[i] - 1, -1, -1):
            k = (k << bits) | int(table[flat[starts[i] + s]])
        want = (k << (64 - int(plen[i]) * bits)) & 0xFFFFFFFFFFFFFFFF if plen[i] else 0
        assert int(keys[i]) == want, i
    sel = torch.from_numpy(np.flatnonzero(ok)).to(dev)
    kk = d_keys[sel].contiguous()
    m = int(ok.sum())
    want_first, want_last = g["count_first"][ok], g["count_last"][ok]
    # count only, 32-bit pairs and 64-bit arrays
    r32 = torch.zeros(2 * m, dtype=torch.int32, device=dev)
    ix.locate_keys_device(m, kk.data_ptr(), 0, r32.data_ptr(), 0, 0, 0, 0, 0, 0, 0)
    f64, l64 = torch.zeros(m, dtype=torch.int64, device=dev), torch.zeros(m, dtype=torch.int64, device=dev)
    ix.locate_keys_device(m, kk.data_ptr(), 0, 0, f64.data_ptr(), l64.data_ptr(), 0, 0, 0, 0, 0)
    torch.cuda.synchronize()
    pairs = r32.cpu().numpy().reshape(m, 2)
    assert np.array_equal(pairs[:, 0], want_first) and np.array_equal(pairs[:, 1], want_last)
    assert np.array_equal(f64.cpu().numpy(), want_first) and np.array_equal(l64.cpu().numpy(), want_last)
    # the whole locate chain on keys against the symbol path on the same patterns
    for mo, _, _ in fx.locate_cases():
        sub_plen, sub_starts = plen[ok], starts[ok]
        noccs_ref, offs_ref = ix.locate_flat(sub_plen, flat, sub_starts, mo)
        cap = int(noccs_ref.sum()) + 8
        noccs = torch.zeros(m, dtype=torch.int32, device=dev)
        ostarts = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        offs = torch.full((cap,), -7, dtype=torch.int64, device=dev)
        total = torch.zeros(2, dtype=torch.int64, device=dev)
        for rep in range(3):     # (the group sums alternate between two sets: several launches in a row must agree)
            ix.locate_keys_device(m, kk.data_ptr(), mo, r32.data_ptr(), 0, 0, noccs.data_ptr(), ostarts.data_ptr(), offs.data_ptr(), cap,
                                  total.data_ptr())
            torch.cuda.synchronize()
            assert total.cpu().tolist() == [int(noccs_ref.sum()), 0], (mo, rep)
            assert np.array_equal(noccs.cpu().numpy(), noccs_ref)
            want_starts = np.concatenate([[0], np.cumsum(noccs_ref.astype(np.int64))])
            assert np.array_equal(ostarts.cpu().numpy(), want_starts)
            assert np.array_equal(offs.cpu().numpy()[:int(noccs_ref.sum())], offs_ref), (mo, rep)
    ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "bytes256", "runs3doc", "b1000"])
def test_budget_sweep_every_plan_answers_the_goldens(fixtures, gpu_ok, name):
    """hbm_budget_bytes from a few KB to more than everything in 28 (12) geometric steps: whatever the planner of api_open.hip makes of a
    budget -- marks of three densities, plain or marked rank units, the text with dense, half-dense (4-byte suffix array + a sampled
    inverse) or sampled arrays, level tables of every depth, context tables, femto's own tables released or kept -- the handle
    holds no more than it may (once the budget covers the block images and the smallest layout) and count, locate, the device chain
    and its row-free form return the goldens.  Several distinct plans must really have been seen."""
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    every = femto_amd.Index(fx.index, device=0, options=dict(hbm_budget_bytes=femto_amd.BUDGET_ALL))
    top = every.structures()["hbm_allocated"]
    every.close()
    plans, within = set(), 0
    steps = 12 if name == "bytes256" else 28          # (bytes256 has the most patterns and clamps: 2.5 s per handle)
    for k in range(steps):
        budget = int(20_000 * (1.6 * top / 20_000) ** (k / (steps - 1.0)))
        ix = femto_amd.Index(fx.index, device=0, options=dict(hbm_budget_bytes=budget))
        st, pi = ix.structures(), ix.pack_info()
        plans.add((ix.rank_mode, pi["sa_full"], pi["isa_full"], pi.get("rank_units", False), pi.get("rank_units_marked", False), pi["ktab_syms"], st["mark_every"],
                   st["image"] == 0, st["text_sa_isa"] > 0, st["context_tables"] > 0, st["char_rank_lines"] > 0))
        within += int(st["hbm_allocated"] <= budget)
        assert st["hbm_allocated"] <= max(budget, top), (budget, st)
        f, l = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f, g["count_first"]) and np.array_equal(l, g["count_last"]), (budget, st)
        for mo, g_noccs, g_offs in fx.locate_cases():
            n_, o_ = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(n_, g_noccs) and np.array_equal(o_, g_offs), (budget, mo, st)
            df, dl, dn, dst, do, dtot = device_locate(ix, plen, flat, starts, mo, len(g_offs) + 16)
            assert dtot == len(g_offs) and np.array_equal(df, g["count_first"]) and np.array_equal(dl, g["count_last"]), (budget, mo)
            assert np.array_equal(dn, g_noccs) and np.array_equal(do, g_offs), (budget, mo, st)
            assert_row_free_equals(ix, plen, flat, starts, mo, g_noccs, g_offs, (name, budget, mo))
        ix.close()
    if any(p_[0] in (3, 4) for p_ in plans):      # (an alphabet of more than 256 characters runs on femto's own tables: one plan)
        assert len(plans) >= 3 and within >= steps // 4, (len(plans), within, sorted(plans))


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_open_with_options(fixtures, gpu_ok, name):
    """femto_amd_open_opts: what is derived is the caller's decision -- a level table of a given depth, none at all, no dense
    arrays, no context tables, a budget of 64 KB (the fixtures are tiny: every optional structure declined) -- and the results never change"""
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    variants = [dict(level_table_syms=2), dict(level_table=0), dict(dense_arrays=0), dict(text=0), dict(context_table=0),
                dict(context2_table=0, context_syms=3), dict(hbm_budget_bytes=1 << 16), dict(char_rank_lines=0), dict(rank_mode=1),
                dict(mark_every=0), dict(tail_min=2, tail_rows=4, tail_row_cost=0), dict(rank_units=0), dict(marks_32bit=0, mark_every=3),
                dict(hbm_budget_bytes=400_000), dict(hbm_budget_bytes=400_000, text=0), dict(level_table_syms=5, rank_units=1, dense_arrays=0),
                dict(context_mid_table=1), dict(context_syms=4, context2_syms=10, context_mid_table=1)]
    for kw in variants:
        ix = femto_amd.Index(fx.index, device=0, options=kw)
        pi = ix.pack_info()
        st = ix.structures()
        if kw.get("rank_units") == 0:
            assert not pi["rank_units"] and st["rank_units"] == 0
        elif name == "acgt48k" and "hbm_budget_bytes" not in kw and kw.get("rank_mode") != 1:
            assert pi["rank_units"] and st["rank_units"] > 0, (kw, st)          # small alphabets get them by default
        if "marks_32bit" in kw:
            assert st["mark_offset_bytes"] == 8 and st["mark_every"] == 3, st
        elif st["marks"]:
            assert st["mark_offset_bytes"] == 4, st
        if kw.get("hbm_budget_bytes", 0) > (1 << 16):
            assert st["hbm_allocated"] <= kw["hbm_budget_bytes"] or st["level_table"] == 0, st
        if "level_table_syms" in kw:
            assert pi["ktab_syms"] == kw["level_table_syms"], (kw, pi)
        if kw.get("level_table") == 0:
            assert not pi["level_table"]
        if kw.get("dense_arrays") == 0 or kw.get("text") == 0:
            assert not pi["sa_full"]
        if kw.get("context_table") == 0:
            assert not pi["context_table"]
        if "context_mid_table" not in kw:
            assert pi["context_mid_syms"] == 0
        if name == "eng2doc" and kw == dict(context_syms=4, context2_syms=10, context_mid_table=1):
            assert pi["context_syms"] == 4 and pi["context2_syms"] == 10 and pi["context_mid_syms"] == 7, pi   # the table half way between
        if "hbm_budget_bytes" in kw:      # (a few hundred KB: no per-character rank lines; the 4-byte suffix array of a tiny fixture may fit its share)
            assert not pi.get("char_rank_lines"), pi
            assert not pi["sa_full"] or (pi["sa_32bit"] and st["hbm_allocated"] <= kw["hbm_budget_bytes"]), (pi, st)
        if kw.get("rank_mode") == 1:
            assert ix.rank_mode == 1
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"]), kw
        for mo, g_noccs, g_offs in fx.locate_cases():
            noccs, offs = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), (kw, mo)
        ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_released_wavelet_lines_come_back(fixtures, gpu_ok, name):
    """A handle with a budget (the default open included) releases femto's wavelet tree as segment lines once the derived
    layouts stand (femto_amd_options_t::wavelet_lines auto) and uploads them again, counted, when a call needs them: leaf
    requests, forward steps, femto_amd_set_rank_mode(1 / 0).  Same goldens before and after; wavelet_lines = 1 keeps them."""
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    kept = femto_amd.Index(fx.index, device=0, options=dict(wavelet_lines=1))
    ix = femto_amd.Index(fx.index, device=0)                    # the default bound: a handle with a budget
    assert ix.structures()["hbm_budget_is_default"] == 1 and ix.rank_mode in (3, 4)
    held0, held_kept = ix.structures()["hbm_allocated"], kept.structures()["hbm_allocated"]
    assert held0 < held_kept, (held0, held_kept)                # the segment lines are gone
    first, last = ix.count_flat(plen, flat, starts)             # the derived layouts do not read them
    assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"])
    assert ix.structures()["hbm_allocated"] == held0
    rows = np.arange(ix.info.total_length, dtype=np.int64)
    ch, occ, off = ix.block_requests(rows)                      # LOCATION leaves read femto's own mark tables: the lines come back
    assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])
    assert held0 < ix.structures()["hbm_allocated"] <= held_kept + 4096
    for mode in (1, 0, ix.rank_mode):
        ix.set_rank_mode(mode)
        f2, l2 = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f2, g["count_first"]) and np.array_equal(l2, g["count_last"]), mode
        for mo, g_noccs, g_offs in fx.locate_cases():
            noccs, offs = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), (mode, mo)
    ix.close()
    ix = femto_amd.Index(fx.index, device=0, options=dict(hbm_budget_bytes=femto_amd.BUDGET_ALL))
    assert ix.structures()["image"] == kept.structures()["image"]          # no budget: nothing released
    ix.close()
    # "bytes this handle may HOLD in all" stays true (round-5 advisor): a budget the segment lines do not fit next to what the handle
    # holds -- CHAR / OCCS requests in modes 3 / 4 never bring them back; LOCATION requests, forward steps and a stay in mode 1 do,
    # and give them up again when the call (the stay) ends
    kch, krow, koff = kept.forward_steps(rows[:64])
    released_seen = 0
    # (a structure set that does not depend on the budget, so that the budget can be placed between it and it + the lines)
    fixed = dict(level_table=0, text=0, rank_units=0, mark_every=5, char_rank_lines=0, context_table=0)
    probe = femto_amd.Index(fx.index, device=0, options=dict(fixed, hbm_budget_bytes=1 << 30))
    t_probe, seg_bytes = probe.structures()["hbm_allocated"], held_kept - held0
    probe.close()
    for budget, opts in ((t_probe + seg_bytes // 2, fixed), (t_probe + 64, fixed), (held0 + seg_bytes // 2, {}), (600_000, {}), (150_000, {})):
        tight = femto_amd.Index(fx.index, device=0, options=dict(opts, hbm_budget_bytes=int(budget)))
        t0 = tight.structures()["hbm_allocated"]
        assert not opts or t0 == t_probe, (t0, t_probe)
        if tight.rank_mode not in (3, 4) or t0 > budget:      # (a budget below the block images themselves: nothing to give up)
            tight.close()
            continue
        ch, occ, off = tight.block_requests(rows, location=False)
        assert off is None and np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and tight.structures()["hbm_allocated"] == t0, budget
        ch, occ, off = tight.block_requests(rows)
        assert np.array_equal(off, g["off"]) and tight.structures()["hbm_allocated"] <= budget, budget
        fch, frow, foff = tight.forward_steps(rows[:64])
        assert np.array_equal(fch, kch) and np.array_equal(frow, krow) and np.array_equal(foff, koff) and tight.structures()["hbm_allocated"] <= budget
        mode34 = tight.rank_mode
        tight.set_rank_mode(1)
        in_mode1 = tight.structures()["hbm_allocated"]
        assert in_mode1 > t0
        f2, l2 = tight.count_flat(plen, flat, starts)
        assert np.array_equal(f2, g["count_first"]) and np.array_equal(l2, g["count_last"])
        tight.set_rank_mode(mode34)
        back = tight.structures()["hbm_allocated"]
        assert back <= budget and back in (t0, in_mode1), (budget, t0, in_mode1, back)
        released_seen += int(in_mode1 > budget and back == t0)
        f2, l2 = tight.count_flat(plen, flat, starts)
        assert np.array_equal(f2, g["count_first"]) and np.array_equal(l2, g["count_last"])
        tight.close()
    assert released_seen >= 1
    kept.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_released_lines_under_concurrent_callers(fixtures, gpu_ok, name):
    """The round-5 advisor's race: femto's own tables come back (an upload under the handle's lock) and go again (over budget) while
    other threads search on the derived layouts and copy the device descriptor for their launches.  Six threads on ONE handle whose
    budget sits between what it holds and that + the lines: two search (count + locate, host and device forms), two ask LOCATION
    leaves (lines up, then released again), one takes forward steps, one count-only leaves (never needs the lines) -- every answer
    equals the goldens, nothing hangs, and the handle ends where it started."""
    import threading
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    kept = femto_amd.Index(fx.index, device=0, options=dict(wavelet_lines=1))
    rows = np.arange(kept.info.total_length, dtype=np.int64)
    kch, krow, koff = kept.forward_steps(rows[:256])
    fixed = dict(level_table=0, text=0, rank_units=0, mark_every=5, char_rank_lines=0, context_table=0)
    probe = femto_amd.Index(fx.index, device=0, options=dict(fixed, hbm_budget_bytes=1 << 30))
    t_probe = probe.structures()["hbm_allocated"]
    probe.close()
    kept.close()
    for budget in (t_probe + 64, 1 << 30):                   # the lines never fit / always fit
        ix = femto_amd.Index(fx.index, device=0, options=dict(fixed, hbm_budget_bytes=int(budget)))
        t0 = ix.structures()["hbm_allocated"]
        assert t0 == t_probe and ix.rank_mode in (3, 4)
        errors = []

        def guard(fn):
            def run():
                try:
                    for _ in range(12):
                        fn()
                except Exception as ex:      # noqa: BLE001
                    errors.append(repr(ex))
            return run

        def search():
            f, l = ix.count_flat(plen, flat, starts)
            assert np.array_equal(f, g["count_first"]) and np.array_equal(l, g["count_last"])
            for mo, g_noccs, g_offs in fx.locate_cases():
                n_, o_ = ix.locate_flat(plen, flat, starts, mo)
                assert np.array_equal(n_, g_noccs) and np.array_equal(o_, g_offs), mo

        def chain():
            for mo, g_noccs, g_offs in fx.locate_cases():
                df, dl, dn, dst, do, dtot = device_locate(ix, plen, flat, starts, mo, len(g_offs) + 16)
                assert dtot == len(g_offs) and np.array_equal(df, g["count_first"]) and np.array_equal(dn, g_noccs) and np.array_equal(do, g_offs), mo

        def leaves():
            ch, occ, off = ix.block_requests(rows)
            assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])

        def forward():
            fch, frow, foff = ix.forward_steps(rows[:256])
            assert np.array_equal(fch, kch) and np.array_equal(frow, krow) and np.array_equal(foff, koff)

        def count_leaves():
            ch, occ, off = ix.block_requests(rows, location=False)
            assert off is None and np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"])

        threads = [threading.Thread(target=guard(fn)) for fn in (search, chain, leaves, leaves, forward, count_leaves)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not any(th.is_alive() for th in threads), "a caller hangs"
        assert not errors, errors[:3]
        held = ix.structures()["hbm_allocated"]
        assert held <= budget and (held == t0 if budget < (1 << 30) else held >= t0), (budget, t0, held)
        search()
        ix.close()


def test_budget_environment_variable_is_validated(fixtures, gpu_ok, monkeypatch):
    """FEMTO_AMD_HBM_BUDGET: bytes with an optional k / M / G / T suffix, or "all" in any case; anything else is not a budget and the
    default bound applies (atoll() used to read "8G" as 8 bytes and "ALL" as 0: every optional structure silently declined)."""
    fx = fixtures("acgt48k")
    for text, want in (("8G", 8 << 30), ("8GiB", 8 << 30), ("512M", 512 << 20), ("3000000", 3_000_000), ("ALL", -1), ("all", -1)):
        monkeypatch.setenv("FEMTO_AMD_HBM_BUDGET", text)
        ix = femto_amd.Index(fx.index, device=0)
        st = ix.structures()
        assert st["hbm_budget"] == want and st["hbm_budget_is_default"] == 0, (text, st)
        ix.close()
    monkeypatch.delenv("FEMTO_AMD_HBM_BUDGET")
    ix = femto_amd.Index(fx.index, device=0)
    dflt = ix.structures()["hbm_budget"]
    ix.close()
    for text in ("8X", "G", "-5", "12 34", ""):
        monkeypatch.setenv("FEMTO_AMD_HBM_BUDGET", text)
        ix = femto_amd.Index(fx.index, device=0)
        st = ix.structures()
        assert st["hbm_budget_is_default"] == 1 and st["hbm_budget"] == dflt and st["level_table_syms"] > 0, (text, st)
        ix.close()


def test_key_table_id(fixtures, gpu_ok):
    """femto_amd_key_table_id: the same for two handles of one index, different for indexes whose characters differ"""
    a, b = femto_amd.Index(fixtures("acgt48k").index, device=0), femto_amd.Index(fixtures("acgt48k").index, device=0, options=dict(hbm_budget_bytes=600_000))
    c = femto_amd.Index(fixtures("eng2doc").index, device=0)
    assert a.key_table_id() == b.key_table_id() != c.key_table_id()
    for ix in (a, b, c):
        ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "runs3doc", "eng2doc"])
def test_level_table_deep_entries_recomputed(fixtures, gpu_ok, monkeypatch, name):
    """The deepest level of the level table stores (first, rows) in 8 bytes; an entry with 2^24 - 1 rows or more stores
    "recompute" and its range is derived from its parent with one ordinary step (ktab2_lookup).  No fixture has 16.7 M rows
    under one K-gram, so the bound is lowered (FEMTO_AMD_KTAB_DEEP_BIG = 1 / 2 / 5): every deepest-level entry with that many
    rows then takes the recomputation, and every result must still be the reference's."""
    fx = fixtures(name)
    g = fx.gold
    plen, flat, starts = fx.patterns
    for big in (1, 2, 5):
        monkeypatch.setenv("FEMTO_AMD_KTAB_DEEP_BIG", str(big))
        for k in (1, 2, 3):
            ix = femto_amd.Index(fx.index, device=0, options=dict(level_table_syms=k))
            assert ix.pack_info()["ktab_syms"] == k
            first, last = ix.count_flat(plen, flat, starts)
            assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"]), (big, k)
            for mo, g_noccs, g_offs in fx.locate_cases():
                noccs, offs = ix.locate_flat(plen, flat, starts, mo)
                assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), (big, k, mo)
            ix.close()


@pytest.mark.parametrize("name,mode", [("acgt48k", 3), ("eng2doc", 4), ("runs3doc", 4), ("eng2doc", "mid")])
def test_pattern_window_every_alignment_and_length(fixtures, gpu_ok, name, mode):
    """The count kernel reads a lane's symbols through aligned 16-byte pieces whose phase depends on the pattern's address
    and length (direct_kernels.hip.hpp): every start address mod 16 bytes x every length 0 .. 150 (one window, its last
    dword, refills), patterns that occur (substrings of the prepared text, some running over SEOF) and patterns spoilt in
    one symbol, laid out in the symbol buffer with caller-chosen gaps -- device entry points against the oracle."""
    import torch
    fx = fixtures(name)
    if mode == "mid":      # ... and with the third context table (context_mid_table: patterns between the two tables' lengths)
        ix = femto_amd.Index(fx.index, device=0, options=dict(two_level_lines=1, rank_mode=4, context_mid_table=1, context_syms=3, context2_syms=9))
        pi = ix.pack_info()
        assert ix.rank_mode == 4 and (pi["context_syms"], pi["context_mid_syms"], pi["context2_syms"]) == (3, 6, 9), pi
    else:
        ix = _open(fx.index, mode)
    o = po.Oracle(fx.index)
    prepared = fx.prepared_text()
    rng = np.random.Generator(np.random.PCG64(4242))
    plen, starts, chunks, pos = [], [], [], 0
    for ln in range(0, 151):
        for phase in range(8):
            gap = (phase - pos) % 8                      # symbol index mod 8 = 16-byte phase of the start address
            chunks.append(rng.integers(5, 261, gap).astype(np.uint16))     # (neighbouring symbols a piece may also hold)
            pos += gap
            s0 = int(rng.integers(0, len(prepared) - ln)) if ln else 0
            p_ = prepared[s0:s0 + ln].copy()
            if ln and rng.random() < 0.3:
                p_[int(rng.integers(0, ln))] = int(rng.choice([3, 5 + 0x41, 5 + 0x7a, 260]))
            plen.append(ln)
            starts.append(pos)
            chunks.append(p_.astype(np.uint16))
            pos += ln
    plen, starts = np.array(plen, dtype=np.int32), np.array(starts, dtype=np.int64)
    flat = np.concatenate(chunks + [np.zeros(8, dtype=np.uint16)])
    n = len(plen)
    of, ol = o.count_flat(plen, flat, starts, threads=8)
    on, oo = o.locate_flat(plen, flat, starts, 5, threads=8)
    assert (ol >= of).sum() > n // 3 and (ol < of).sum() > n // 10
    dev = "cuda:0"
    base = torch.zeros(len(flat) + 8, dtype=torch.int16, device=dev)
    for shift in (0, 3):                                  # the buffer itself at two different alignments
        d_flat = base[shift:shift + len(flat)]
        d_flat.copy_(torch.from_numpy(flat.view(np.int16)))
        d_plen, d_starts = torch.from_numpy(plen).to(dev), torch.from_numpy(starts).to(dev)
        f, l = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        ix.count_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), f.data_ptr(), l.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(f.cpu().numpy(), of) and np.array_equal(l.cpu().numpy(), ol), shift
        cap = int(on.sum()) + 8
        noccs = torch.zeros(n, dtype=torch.int32, device=dev)
        ostarts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        offs = torch.full((cap,), -7, dtype=torch.int64, device=dev)
        total = torch.zeros(2, dtype=torch.int64, device=dev)
        ix.locate_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), 5, f.data_ptr(), l.data_ptr(), noccs.data_ptr(),
                         ostarts.data_ptr(), offs.data_ptr(), cap, total.data_ptr())
        torch.cuda.synchronize()
        assert total.cpu().tolist() == [int(on.sum()), 0]
        assert np.array_equal(noccs.cpu().numpy(), on)
        assert np.array_equal(offs.cpu().numpy()[:int(on.sum())], oo), shift
    ix.close()

"""`-m gpu` parity tests: the HIP path (through the C ABI, femto_amd/libfemto_amd.so) against the
oracle and the committed golden vectors of the genuine reference.  Bit-exact (integer work)."""
import ctypes as C
import os

import numpy as np
import pytest

import femto_amd
from conftest import INDEX_FIXTURES
from femto_amd import textgen as tg
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ok():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return True


# 3: packed small-alphabet lines (default when the index has <= 8 characters); 4: two-level 16-ary lines (default for
# 9..256 characters); 1: lane per query on femto's wavelet tree (default otherwise); 2: flattened persistent lanes;
# 0: wavefront-per-query raw walk
MODES = [3, 4, 1, 0]


def _torchrun(nproc, script_and_args, env, cwd=None, attempts=2):
    """python -m torch.distributed.run on 127.0.0.1 with a free port; one retry (a port can be taken between probing and use)"""
    import socket
    import subprocess
    import sys
    out = None
    for _ in range(attempts):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_and_args,
                             env=env, capture_output=True, text=True, timeout=600, cwd=cwd)
        if out.returncode == 0:
            break
    return out


def _open(path, mode=None):
    """open on GPU 0; mode 4 is built for small alphabets too (FEMTO_AMD_PACK2=1) so that every fixture exercises it"""
    old = os.environ.get("FEMTO_AMD_PACK2")
    os.environ["FEMTO_AMD_PACK2"] = "1"
    try:
        ix = femto_amd.Index(path, device=0)
    finally:
        if old is None:
            del os.environ["FEMTO_AMD_PACK2"]
        else:
            os.environ["FEMTO_AMD_PACK2"] = old
    if mode is not None:
        _set_mode(ix, mode)
    return ix


def _set_mode(ix, mode):
    if mode == 3 and not ix.pack_info()["available"]:
        ix.close()
        pytest.skip("more than 8 distinct characters: no packed lines for this index")
    if mode == 4 and not ix.pack_info()["available2"]:
        ix.close()
        pytest.skip("more than 256 distinct characters: no two-level lines for this index")
    ix.set_rank_mode(mode)
    assert ix.rank_mode == mode


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


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "chunks2doc"])
def test_multi_device_handle_shards_host_batches(fixtures, gpu_ok, name):
    """femto_amd_open_multi: one handle over several GPUs of the process; every host-pointer batch call splits into
    contiguous shards, one host thread per replica, results straight into the caller's arrays.  The box has one GPU, so
    the three replicas share it -- the sharding, the merging of located offsets and the error path are what is tested."""
    fx = fixtures(name)
    g = fx.gold
    ix = femto_amd.Index(fx.index, devices=[0, 0, 0])
    assert femto_amd.lib().femto_amd_device_count(ix.handle) == 3
    plen, flat, starts = fx.patterns
    first, last = ix.count_flat(plen, flat, starts)
    assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"])
    for mo, g_noccs, g_offs in fx.locate_cases():
        noccs, offs = ix.locate_flat(plen, flat, starts, mo)
        assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), mo
        noccs2, offs2 = ix.locate_flat_two_call(plen, flat, starts, mo)
        assert np.array_equal(noccs2, g_noccs) and np.array_equal(offs2, g_offs), mo
    # the reference's own calling convention (alpha_t**, callee-malloc'd offsets[i])
    n = len(plen)
    L = femto_amd.lib()
    pats = [np.ascontiguousarray(flat[starts[i]:starts[i] + plen[i]]) for i in range(n)]
    parr = (C.c_void_p * n)(*[p.ctypes.data if len(p) else None for p in pats])
    pl = plen.astype(np.int32)
    f2 = np.zeros(n, dtype=np.int64)
    l2 = np.zeros(n, dtype=np.int64)
    assert L.femto_amd_parallel_count(ix.handle, n, pl.ctypes.data, parr, f2.ctypes.data, l2.ctypes.data) == 0
    assert np.array_equal(f2, g["count_first"]) and np.array_equal(l2, g["count_last"])
    noccs = np.zeros(n, dtype=np.int32)
    offs = (C.POINTER(C.c_int64) * n)()
    assert L.femto_amd_parallel_locate(ix.handle, n, pl.ctypes.data, parr, 7, noccs.ctypes.data, offs) == 0
    assert np.array_equal(noccs, g["loc7_noccs"])
    got = []
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for i in range(n):
        if noccs[i]:
            got.extend(offs[i][j] for j in range(noccs[i]))
            libc.free(offs[i])
    assert np.array_equal(np.array(got, dtype=np.int64), g["loc7_offs"])
    rows = int(ix.info.total_length)
    single = femto_amd.Index(fx.index, device=0)
    assert np.array_equal(ix.locate_range(0, rows - 1), single.locate_range(0, rows - 1))
    single.close()
    # a device-pointer call has no meaning on such a handle
    with pytest.raises(femto_amd.FemtoAmdError):
        ix.count_device(1, 8, 8, 8, 8, 8)
    # a bad pattern in one shard fails the whole call with that shard's error
    bad = flat.copy()
    bad[int(starts[n - 1])] = 300 if plen[n - 1] else bad[0]
    if plen[n - 1]:
        with pytest.raises(femto_amd.FemtoAmdError) as ei:
            ix.count_flat(plen, bad, starts)
        assert ei.value.code == 3
    ix.close()


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "bytes256"])
def test_striped_index_over_devices(fixtures, gpu_ok, name):
    """femto_amd_open_multi_striped: every big array is one address range whose pages are spread over the listed GPUs
    (HIP virtual memory management), the small tables are copied per GPU, the kernels are unchanged.  The box has one GPU,
    so the three stripes and the two views live on it -- allocation, mapping, the per-stripe copies / fills and the views'
    table copies are what is tested; every kernel family must still reproduce the goldens through views."""
    fx = fixtures(name)
    g = fx.gold
    ix = femto_amd.Index(fx.index, devices=[0, 0, 0], striped=True)
    plen, flat, starts = fx.patterns
    for mode in (None, 1):
        if mode is not None:
            ix.set_rank_mode(mode)
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"]), mode
        for mo, g_noccs, g_offs in fx.locate_cases():
            noccs, offs = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), (mode, mo)
    rows = int(ix.info.total_length)
    single = femto_amd.Index(fx.index, device=0)
    assert np.array_equal(ix.locate_range(0, rows - 1), single.locate_range(0, rows - 1))
    ch, occ, off = ix.block_requests(np.arange(rows, dtype=np.int64))
    assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])
    single.close()
    ix.close()


def test_comm_gather_one_rank(fixtures, gpu_ok):
    """femto_amd_comm_*: RCCL is loaded on first use; a communicator of one rank gathers to itself (the N > 1 exchange is
    the same grouped ncclSend / ncclRecv batch, which needs N GPUs: bench.py --gather native on the multi-GPU node)."""
    import torch
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=0)
    ix.comm_init(femto_amd.Index.comm_unique_id(), 1, 0)
    src = torch.arange(1000, dtype=torch.int64, device="cuda:0")
    dst = torch.zeros(1000, dtype=torch.int64, device="cuda:0")
    ix.comm_gather(src.data_ptr(), dst.data_ptr(), 8000, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    ix.close()


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


def test_full_size_1gib_properties(tmp_path, gpu_ok):
    """BASELINE configs[1] at FULL size (1 GiB random-ACGT text, reference default parameters), checked through
    size-independent properties plus an oracle spot check:
      * every 20-mer sampled from the text is found, and every located offset really is an occurrence
        (text[off : off+20] == pattern), offsets of a pattern are distinct, noccs == count (below the clamp);
      * locating one whole bucket-aligned row range returns distinct offsets whose preceding characters
        are the L column (the LF invariant) -- i.e. SA and BWT agree;
      * 3 000 random + sampled patterns agree bit-for-bit with the oracle (count and locate)."""
    text = tg.t_acgt(1 << 30, 424242)
    path = str(tmp_path / "acgt1g")
    femto_amd.build_index(path, [text], params=None, infos=["full"], device=0)
    ix = femto_amd.Index(path, device=0)
    assert ix.info.total_length == (1 << 30) + 1 and ix.info.number_of_blocks == 9 and ix.info.total_buckets == 1025
    npat = 1_000_000
    plen, flat = tg.p_hit(20, 20, npat, 11, text)
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    cnt = last - first + 1
    assert (cnt >= 1).all()
    noccs, offs = ix.locate_flat(plen, flat, starts, 100)
    assert np.array_equal(noccs, np.minimum(cnt, np.where(cnt - 1 > 100, 100, cnt)))
    owner = np.repeat(np.arange(npat), noccs)
    pat_bytes = (flat.reshape(npat, 20) - 5).astype(np.uint8)
    for k in range(20):       # column-wise compare keeps memory bounded
        assert np.array_equal(text[offs + k], pat_bytes[owner, k]), k
    key = owner.astype(np.int64) * (1 << 31) + offs
    assert len(np.unique(key)) == len(key)
    # LF / LF^-1 consistency on rows inside the 'C' range
    f1, l1 = ix.count([tg.to_alpha(np.frombuffer(b"C", dtype=np.uint8))])
    r0 = int(f1[0]) + (1 << 27) + 12345          # one million consecutive rows of the 'C' range, crossing a block boundary
    rows = np.arange(r0, r0 + 1_000_000, dtype=np.int64)
    assert f1[0] <= r0 and r0 + 1_000_000 - 1 <= l1[0]
    # offsets of rows r0.. via LF^-1: F[row] == 'C' and text[SA[row]] == 'C'
    fch, frow, _ = ix.forward_steps(rows[:100000])
    assert (fch == 5 + ord("C")).all()
    lch, _, _ = ix.block_requests(frow)
    assert (lch == 5 + ord("C")).all()                      # L[LF^-1(row)] == F[row]
    # oracle spot check
    o = po.Oracle(path)
    rp, rf = tg.p_rand(20, 1500, 3)
    p2 = np.concatenate([rp, plen[:1500]])
    f2 = np.concatenate([rf, flat[:1500 * 20]])
    s2 = tg.starts_of(p2)
    gf, gl = ix.count_flat(p2, f2, s2)
    of, ol = o.count_flat(p2, f2, s2, threads=16)
    assert np.array_equal(gf, of) and np.array_equal(gl, ol)
    gn, go = ix.locate_flat(p2, f2, s2, 100)
    on, oo = o.locate_flat(p2, f2, s2, 100, threads=16)
    assert np.array_equal(gn, on) and np.array_equal(go, oo)
    # the packed lines (default here) and the wavelet path agree on the whole million-pattern batch and on
    # two million leaf requests spread over all rows
    assert ix.rank_mode == 3
    rows2 = np.random.Generator(np.random.PCG64(17)).integers(0, ix.info.total_length, 2_000_000).astype(np.int64)
    leaf3 = ix.block_requests(rows2)
    ix.set_rank_mode(1)
    first1, last1 = ix.count_flat(plen, flat, starts)
    assert np.array_equal(first1, first) and np.array_equal(last1, last)
    noccs1, offs1 = ix.locate_flat(plen, flat, starts, 100)
    assert np.array_equal(noccs1, noccs) and np.array_equal(offs1, offs)
    leaf1 = ix.block_requests(rows2)
    for a, b in zip(leaf3, leaf1):
        assert np.array_equal(a, b)
    # The headline's own regime (round-3 verdict, task 6): 1 M RANDOM 20-mers -- four of five die inside the level table
    # (K = 16), and the (first, last) of a dead range must be the values of the step that emptied it (server.c:832-936).
    # Mode 3 (table + rank units / packed lines) against mode 1 (femto's wavelet tree, no table) on all of them, against the
    # oracle on 50 000, dead ranges compared explicitly; then the same under the footprint-bounded option set.
    rplen, rflat = tg.p_rand(20, 1_000_000, 77)
    rstarts = tg.starts_of(rplen)
    rf1, rl1 = ix.count_flat(rplen, rflat, rstarts)                   # (mode 1 is set)
    rn1, ro1 = ix.locate_flat(rplen, rflat, rstarts, 100)
    ix.set_rank_mode(3)
    assert ix.pack_info()["ktab_syms"] == 16 and ix.pack_info()["rank_units"]
    rf3, rl3 = ix.count_flat(rplen, rflat, rstarts)
    rn3, ro3 = ix.locate_flat(rplen, rflat, rstarts, 100)
    dead = rl3 < rf3
    assert 0.99 < dead.mean() < 1.0 and (rl3[dead] == rf3[dead] - 1).all()
    assert np.array_equal(rf3[dead], rf1[dead]) and np.array_equal(rl3[dead], rl1[dead])        # the emptying step's values
    assert np.array_equal(rf3, rf1) and np.array_equal(rl3, rl1) and np.array_equal(rn3, rn1) and np.array_equal(ro3, ro1)
    m = 50_000
    of, ol = o.count_flat(rplen[:m], rflat, rstarts[:m], threads=32)
    assert np.array_equal(of, rf3[:m]) and np.array_equal(ol, rl3[:m])
    on, oo = o.locate_flat(rplen[:m], rflat, rstarts[:m], 100, threads=32)
    assert np.array_equal(on, rn3[:m]) and np.array_equal(oo, ro3[:int(rn3[:m].sum())])
    ix.close()
    # footprint-bounded open (hbm_budget_bytes = 4 x text): rank units + packed lines + sampled marks + the level table the
    # rest pays for, no dense arrays, no text -- the handle holds what it was allowed, and answers identically
    bx = femto_amd.Index(path, device=0, options=dict(hbm_budget_bytes=4 << 30))
    st = bx.structures()
    assert st["hbm_allocated"] <= (4 << 30) and st["rank_units"] > 0 and st["text_sa_isa"] == 0 and 12 <= st["level_table_syms"] <= 14, st
    assert not bx.pack_info()["sa_full"] and bx.rank_mode == 3
    bf, bl = bx.count_flat(rplen, rflat, rstarts)
    assert np.array_equal(bf, rf3) and np.array_equal(bl, rl3)
    bn, bo = bx.locate_flat(rplen, rflat, rstarts, 100)
    assert np.array_equal(bn, rn3) and np.array_equal(bo, ro3)
    bf, bl = bx.count_flat(plen, flat, starts)                        # the sampled batch: every step runs, every row is walked to a mark
    assert np.array_equal(bf, first) and np.array_equal(bl, last)
    bn, bo = bx.locate_flat(plen, flat, starts, 100)
    assert np.array_equal(bn, noccs) and np.array_equal(bo, offs)
    bx.close()
    # the same index with its big arrays striped over "three GPUs" (all stripes on this one): same kernels, same answers
    sx = femto_amd.Index(path, devices=[0, 0, 0], striped=True)
    fs, ls = sx.count_flat(plen, flat, starts)
    assert np.array_equal(fs, first) and np.array_equal(ls, last)
    ns, os_ = sx.locate_flat(plen, flat, starts, 100)
    assert np.array_equal(ns, noccs) and np.array_equal(os_, offs)
    sx.close()


def test_full_size_text96_properties(tmp_path, gpu_ok):
    """BASELINE configs[2] at FULL size (1 GiB sigma~96 text, reference default parameters) on the two-level lines
    (mode 4): every sampled pattern of length 8..64 is found; every located offset really is an occurrence; the
    wavelet path (mode 1) agrees on a 200 k-pattern batch and on a million leaf requests; oracle spot check."""
    text = tg.t_eng_torch(1 << 30, 515, "cuda:0")
    path = str(tmp_path / "eng1g")
    femto_amd.build_index(path, [text], params=None, infos=["full"], device=0)
    ix = femto_amd.Index(path, device=0)
    assert ix.info.total_length == (1 << 30) + 1 and ix.rank_mode == 4
    npat = 200_000
    plen, flat = tg.p_hit(8, 64, npat, 12, text)
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    cnt = last - first + 1
    assert (cnt >= 1).all()
    noccs, offs = ix.locate_flat(plen, flat, starts, 20)
    assert np.array_equal(noccs, np.minimum(cnt, np.where(cnt - 1 > 20, 20, cnt)))
    owner = np.repeat(np.arange(npat), noccs)
    for k in range(8):                                   # the first 8 symbols of every located occurrence
        assert np.array_equal(text[offs + k].astype(np.uint16) + 5, flat[starts[owner] + k]), k
    tail = plen[owner] - 1                               # ... and the last one
    assert np.array_equal(text[offs + tail].astype(np.uint16) + 5, flat[starts[owner] + tail])
    rows = np.random.Generator(np.random.PCG64(3)).integers(0, ix.info.total_length, 1_000_000).astype(np.int64)
    leaf4 = ix.block_requests(rows)
    ix.set_rank_mode(1)
    f1, l1 = ix.count_flat(plen, flat, starts)
    assert np.array_equal(f1, first) and np.array_equal(l1, last)
    n1, o1 = ix.locate_flat(plen, flat, starts, 20)
    assert np.array_equal(n1, noccs) and np.array_equal(o1, offs)
    for a, b in zip(leaf4, ix.block_requests(rows)):
        assert np.array_equal(a, b)
    ix.set_rank_mode(4)
    o = po.Oracle(path)
    m = 2000
    of, ol = o.count_flat(plen[:m], flat, starts[:m], threads=16)
    assert np.array_equal(of, first[:m]) and np.array_equal(ol, last[:m])
    on, oo = o.locate_flat(plen[:m], flat, starts[:m], 20, threads=16)
    assert np.array_equal(on, noccs[:m]) and np.array_equal(oo, offs[:int(noccs[:m].sum())])
    # The MISS paths at full size (round-4 verdict, task 1): patterns that mostly do NOT occur -- 100 k uniform over the text's
    # alphabet, lengths 8..64 (they leave the hashed context tables on a 9- / 16-gram the text does not hold, and the table has to
    # hand back so that the emptying step's (first, last) comes out, direct_kernels.hip.hpp), and 100 k sampled patterns with ONE
    # byte substituted (they miss or hit the tables depending on where the substitution falls, and die in the rank steps or the
    # text tail).  max_occs 100 as benchmarked.  Mode 4 against mode 1 (femto's own wavelet tree: no tables at all) on all of
    # them, against the oracle on 60 000, the (first, last) of dead ranges compared explicitly.
    rng = np.random.Generator(np.random.PCG64(99))
    alphabet = np.flatnonzero(np.bincount(text[:1 << 26], minlength=256)).astype(np.uint16) + 5
    assert 90 <= len(alphabet) <= 100
    nmiss = 100_000
    rlen = rng.integers(8, 65, nmiss).astype(np.int32)
    rflat = alphabet[rng.integers(0, len(alphabet), int(rlen.sum()))].astype(np.uint16)
    mlen, mflat = tg.p_hit(8, 64, nmiss, 13, text)
    mstarts = tg.starts_of(mlen)
    at = mstarts + rng.integers(0, 1 << 30, nmiss) % mlen
    mflat = mflat.copy()
    mflat[at] = alphabet[rng.integers(0, len(alphabet), nmiss)]
    qlen = np.concatenate([rlen, mlen, plen[:50_000]])
    qflat = np.concatenate([rflat, mflat, flat[:int(starts[50_000])]])
    qstarts = tg.starts_of(qlen)
    f4, l4 = ix.count_flat(qlen, qflat, qstarts)
    n4, o4 = ix.locate_flat(qlen, qflat, qstarts, 100)
    dead = l4 < f4
    assert 0.5 < dead.mean() < 0.85 and dead[:nmiss].mean() > 0.99 and 0.5 < dead[nmiss:2 * nmiss].mean() < 1.0
    c4 = l4 - f4 + 1
    assert np.array_equal(n4, np.where(dead, 0, np.minimum(c4, np.where(c4 - 1 > 100, 100, c4))))
    ix.set_rank_mode(1)
    f1, l1 = ix.count_flat(qlen, qflat, qstarts)
    n1, o1 = ix.locate_flat(qlen, qflat, qstarts, 100)
    assert np.array_equal(f4[dead], f1[dead]) and np.array_equal(l4[dead], l1[dead])          # the emptying step's values
    assert np.array_equal(f4, f1) and np.array_equal(l4, l1) and np.array_equal(n4, n1) and np.array_equal(o4, o1)
    ix.set_rank_mode(4)
    pick = np.concatenate([np.arange(0, 20_000), np.arange(nmiss, nmiss + 20_000), np.arange(2 * nmiss, 2 * nmiss + 20_000)])
    sub_len = qlen[pick]
    sub_flat = np.concatenate([qflat[qstarts[i]:qstarts[i] + qlen[i]] for i in pick])
    sub_starts = tg.starts_of(sub_len)
    of, ol = o.count_flat(sub_len, sub_flat, sub_starts, threads=32)
    assert np.array_equal(of, f4[pick]) and np.array_equal(ol, l4[pick])
    on, oo = o.locate_flat(sub_len, sub_flat, sub_starts, 100, threads=32)
    o_starts = np.concatenate([[0], np.cumsum(n4)])
    want = np.concatenate([o4[o_starts[i]:o_starts[i + 1]] for i in pick])
    assert np.array_equal(on, n4[pick]) and np.array_equal(oo, want)
    ix.close()


def test_full_size_8gib_properties(tmp_path, gpu_ok):
    """BASELINE configs[4]'s index at FULL size: 8 GiB random-ACGT text (8 589 934 593 rows, 65 data blocks, 64-bit rows
    everywhere), built here by the partitioned 64-bit suffix sorter, opened (a) replicated on the GPU and (b) range-split
    in two parts.  Size-independent properties plus an oracle spot check:
      * every sampled 20-mer is found and every located offset really is an occurrence;
      * the packed lines (default) and the wavelet path (mode 1) agree; the two-part range-split handle agrees with both;
      * 20 000 random + sampled patterns agree bit-for-bit with the oracle (count and locate)."""
    import shutil
    free_disk = shutil.disk_usage(str(tmp_path)).free
    if free_disk < 12 * (1 << 30):
        why = "BASELINE configs[4] (8 GiB index) NOT TESTED on this box: %.1f GB of scratch disk free, ~10 GB needed" % (free_disk / 1e9)
        print("\n*** " + why + " ***", flush=True)
        pytest.skip(why)
    try:
        import psutil
        avail = psutil.virtual_memory().available
        if avail < 200 * (1 << 30):
            why = "BASELINE configs[4] (8 GiB index) NOT TESTED on this box: %.0f GB of host memory available, ~200 GB needed for the text and its suffix array" % (avail / 1e9)
            print("\n*** " + why + " ***", flush=True)
            pytest.skip(why)
    except ImportError:
        pass
    n = 1 << 33
    text = tg.t_acgt(n, 808)
    path = str(tmp_path / "acgt8g")
    femto_amd.build_index(path, [text], params=None, infos=["full8"], device=0)
    ix = femto_amd.Index(path, device=0)
    assert ix.info.total_length == n + 1 and ix.info.number_of_blocks == 65 and ix.info.total_buckets == 8193
    assert ix.info.text_size_bits == 34 and ix.rank_mode == 3
    npat = 200_000
    plen, flat = tg.p_hit(20, 20, npat, 21, text)
    starts = tg.starts_of(plen)
    first, last = ix.count_flat(plen, flat, starts)
    cnt = last - first + 1
    assert (cnt >= 1).all() and last.max() > (1 << 32)          # rows beyond 32 bits are really in play
    noccs, offs = ix.locate_flat(plen, flat, starts, 100)
    assert np.array_equal(noccs, np.minimum(cnt, np.where(cnt - 1 > 100, 100, cnt)))
    assert offs.max() > (1 << 32)
    owner = np.repeat(np.arange(npat), noccs)
    pat_bytes = (flat.reshape(npat, 20) - 5).astype(np.uint8)
    for k in range(20):
        assert np.array_equal(text[offs + k], pat_bytes[owner, k]), k
    # oracle spot check (random + sampled)
    o = po.Oracle(path)
    rp, rf = tg.p_rand(20, 10_000, 5)                    # (round-4 verdict: >= 20 k patterns against the oracle at this size)
    p2 = np.concatenate([rp, plen[:10_000]])
    f2 = np.concatenate([rf, flat[:10_000 * 20]])
    s2 = tg.starts_of(p2)
    gf, gl = ix.count_flat(p2, f2, s2)
    of, ol = o.count_flat(p2, f2, s2, threads=16)
    assert np.array_equal(gf, of) and np.array_equal(gl, ol)
    gn, go = ix.locate_flat(p2, f2, s2, 100)
    on, oo = o.locate_flat(p2, f2, s2, 100, threads=16)
    assert np.array_equal(gn, on) and np.array_equal(go, oo)
    # wavelet path on the same handle
    m = 50_000
    ix.set_rank_mode(1)
    f1, l1 = ix.count_flat(plen[:m], flat, starts[:m])
    assert np.array_equal(f1, first[:m]) and np.array_equal(l1, last[:m])
    n1, o1 = ix.locate_flat(plen[:m], flat, starts[:m], 100)
    assert np.array_equal(n1, noccs[:m]) and np.array_equal(o1, offs[:int(noccs[:m].sum())])
    ix.close()
    del text
    # range-split in two parts (both on this GPU): part p keeps blocks [65p/2, 65(p+1)/2) and reads the rest from its peer
    parts = [femto_amd.Index(path, device=0, part=p, nparts=2) for p in range(2)]
    for a in parts:
        for b in parts:
            if a is not b:
                a.split_attach_local(b)
    for a in parts:
        a.split_commit()
    for a in parts:
        fs, ls = a.count_flat(plen[:m], flat, starts[:m])
        assert np.array_equal(fs, first[:m]) and np.array_equal(ls, last[:m])
        ns, os_ = a.locate_flat(plen[:m], flat, starts[:m], 100)
        assert np.array_equal(ns, noccs[:m]) and np.array_equal(os_, offs[:int(noccs[:m].sum())])
    for a in parts:
        a.close()


def test_multiquery_cpp_tool(fixtures, tmp_path, gpu_ok):
    """femto_amd_multiquery (C++ host over the C ABI, femto_multiquery's counterpart): Pizza&Chili query file on
    stdin, -count / -locate [max]; dumped results must equal the oracle's."""
    import subprocess
    from femto_amd import build as b
    tool = b.build_tools()
    fx = fixtures("eng2doc")
    text = np.concatenate(fx.docs)
    rng = np.random.Generator(np.random.PCG64(1))
    n, m = 500, 6
    startpos = rng.integers(0, len(text) - m, n)
    pats = np.stack([text[s0:s0 + m] for s0 in startpos])
    qfile = f"# number={n} length={m} file=test forbidden=\n".encode() + pats.tobytes()
    o = po.Oracle(fx.index)
    alpha = [tg.to_alpha(p) for p in pats]
    of, ol = o.count(alpha)
    dump = str(tmp_path / "c.bin")
    r = subprocess.run([tool, fx.index, "-count", "--dump", dump], input=qfile, capture_output=True, check=True)
    assert f"Counted {int((ol - of + 1).sum())} results".encode() in r.stdout
    got = np.fromfile(dump, dtype=np.int64)
    assert np.array_equal(got[:n], of) and np.array_equal(got[n:], ol)
    on, oo = o.locate(alpha, 5)
    r = subprocess.run([tool, fx.index, "-locate", "5", "--dump", dump], input=qfile, capture_output=True, check=True)
    raw = open(dump, "rb").read()
    assert np.array_equal(np.frombuffer(raw, dtype=np.int32, count=n), on)
    assert np.array_equal(np.frombuffer(raw, dtype=np.int64, offset=4 * n), oo)


# ---- range-split index (femto_amd_open_split): blocks partitioned over parts, remote slices mapped ----------

def _open_split_local(path, nparts):
    parts = [femto_amd.Index(path, device=0, part=p, nparts=nparts) for p in range(nparts)]
    for a in parts:
        for b in parts:
            if a is not b:
                a.split_attach_local(b)
    for a in parts:
        a.split_commit()
    return parts


@pytest.mark.parametrize("nparts", [2, 3, 8])
@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_range_split_matches_reference_goldens(fixtures, gpu_ok, name, nparts):
    """Every part of a range-split index (its own blocks in its own allocation, the others' reached through
    rebased offsets) answers leaf requests, count and locate exactly like the reference."""
    fx = fixtures(name)
    g = fx.gold
    parts = _open_split_local(fx.index, nparts)
    nb = parts[0].info.number_of_blocks
    infos = [p.split_info() for p in parts]
    assert sum(1 for i in infos if i["seg_bytes"] > 0) == min(nb, nparts)
    whole = femto_amd.Index(fx.index, device=-1)
    assert sum(i["image_bytes"] for i in infos) <= whole.info.image_bytes
    plen, flat, starts = fx.patterns
    n = parts[0].info.total_length
    rows = np.arange(n, dtype=np.int64)
    for ix in parts:
        ch, occ, off = ix.block_requests(rows)
        assert np.array_equal(ch, g["L"])
        assert np.array_equal(occ, g["occ"])
        assert np.array_equal(off, g["off"])
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"])
        assert np.array_equal(last, g["count_last"])
        for mo, noccs, offs in fx.locate_cases():
            k, got = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(k, noccs), mo
            assert np.array_equal(got, offs), mo
    for ix in parts:
        ix.close()


def test_range_split_needs_every_part(fixtures, gpu_ok):
    fx = fixtures("acgt48k")
    a = femto_amd.Index(fx.index, device=0, part=0, nparts=2)
    plen, flat, starts = fx.patterns
    with pytest.raises(femto_amd.FemtoAmdError) as ei:
        a.count_flat(plen, flat, starts)
    assert ei.value.code == 6   # ERR_INVALID
    with pytest.raises(femto_amd.FemtoAmdError):
        a.split_commit()
    b = femto_amd.Index(fx.index, device=0, part=1, nparts=2)
    a.split_attach_local(b)
    a.split_commit()
    with pytest.raises(femto_amd.FemtoAmdError):
        a.set_rank_mode(0)
    with pytest.raises(femto_amd.FemtoAmdError):
        a.forward_steps(np.arange(4, dtype=np.int64))
    first, last = a.count_flat(plen, flat, starts)
    assert np.array_equal(first, fx.gold["count_first"])
    a.close()
    b.close()


def test_range_split_across_processes(fixtures, gpu_ok, tmp_path):
    """Two PROCESSES (one rank each, both on this box's single GPU): hipIpc handles travel through
    torch.distributed, each rank maps the other's slices and answers the whole golden batch."""
    fx = fixtures("acgt48k")
    script = os.path.join(os.path.dirname(__file__), "split_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _torchrun(2, [script, fx.index, os.path.join(os.path.dirname(__file__), "golden", "acgt48k.npz"), str(tmp_path)], env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(2):
        assert (tmp_path / f"ok{r}").exists()


def test_striped_index_across_processes(fixtures, gpu_ok, tmp_path):
    """Two PROCESSES (both on this box's single GPU) share ONE striped index: rank 0 derives it
    (femto_amd_open_multi_striped, two stripes) and serves every stripe as a file descriptor over a Unix socket
    (femto_amd_striped_serve); rank 1 maps them at the same addresses (femto_amd_open_striped_client) and answers the
    golden batches on the packed lines (DNA fixture) and on the two-level lines + context tables (byte fixture) -- the
    fast paths, not the wavelet path of the IPC range-split -- and through the enqueue-only device chain."""
    script = os.path.join(os.path.dirname(__file__), "striped_worker.py")
    gold = os.path.join(os.path.dirname(__file__), "golden")
    args = [str(tmp_path)]
    for name, mode in (("acgt48k", 3), ("eng2doc", 4)):
        args += [fixtures(name).index, os.path.join(gold, f"{name}.npz"), str(mode)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _torchrun(2, [script] + args, env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(2):
        assert (tmp_path / f"ok{r}").exists()


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


@pytest.mark.parametrize("layout", ["replicated", "striped"])
def test_bench_two_ranks_control_flow(tmp_path, gpu_ok, layout):
    """bench.py's N > 1 path (rank 0 builds, everybody opens, sharded steps, double-buffered gather of the narrowed
    ranges, max-over-ranks timing, one JSON line from rank 0) with two ranks sharing this box's GPU and the gather routed
    through gloo -- the control flow the driver runs with RCCL on 2/4/8 GPUs.  (Random 20-mers match next to nothing, so
    the list of patterns with 255 matches or more stays empty here; tests/test_distributed.py fills it.)"""
    import json
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, FEMTO_AMD_BENCH_BACKEND="gloo", FEMTO_AMD_BENCH_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    out = _torchrun(2, [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--text-log2", "22",
                        "--npats", "200000", "--cpu-sample", "2000", "--layout", layout], env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert line["config"]["patterns_per_gpu"] == 200000
    # the buffers that arrived on rank 0 (one byte per match count + offsets, femto_amd_pack_counts_device) decoded, and rank
    # 0's own slot equalled its local results
    assert line["config"]["gathered_results_verified"] is True
    assert line["config"]["parallelism"].startswith("striped index" if layout == "striped" else "replicated index")


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
               dict(dense_arrays=0, text=0, rank_units=0), dict(hbm_budget_bytes=600_000), dict(hbm_budget_bytes=150_000)):
        ix = femto_amd.Index(fx.index, device=0, options=kw)
        assert "hbm_budget_bytes" in kw or not ix.pack_info()["sa_full"]
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
                    assert np.array_equal(offs.cpu().numpy()[:min(cap, tot)], g_offs[:min(cap, tot)]), (kw, mo, cap, rep)
        ix.close()


def test_bench_eight_ranks_dry_run(tmp_path, gpu_ok):
    """The argument path of the driver's 8-GPU scaling run (`bench.py --gpus 8` under torch.distributed.run), dry: eight
    ranks sharing this box's GPU, the gather through gloo, a 16 MiB text.  It must finish under the watchdogs and print ONE
    headline line with eight `config.per_rank` entries (search / gather-stall times, payload bytes, world size seen) -- what
    makes the first hardware run self-explaining.  Never a measurement."""
    import json
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, FEMTO_AMD_BENCH_BACKEND="gloo", FEMTO_AMD_BENCH_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    out = _torchrun(8, [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--text-log2", "24",
                        "--npats", "100000", "--cpu-sample", "2000"], env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    line = lines[-1]
    assert "metric" in line and line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    pr = line["config"]["per_rank"]
    assert len(pr) == 8 and sorted(r["rank"] for r in pr) == list(range(8)) and all(r["world_size_seen"] == 8 for r in pr)
    assert line["config"]["gathered_results_verified"] is True and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert all("extra" in ln for ln in lines[:-1])          # whatever precedes the headline is an `extra` line


@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc"])
def test_keys_device_path_equals_symbol_path(fixtures, gpu_ok, name):
    """femto_amd_pack_keys_device + femto_amd_locate_keys_device: patterns as 64-bit keys, ranges as int32 pairs -- the same
    (first, last), clamped row counts, out_starts and located offsets as the symbol entry points and the reference's goldens,
    for every pattern a key describes; the others are counted in *d_bad"""
    import torch
    fx = fixtures(name)
    g = fx.gold
    ix = femto_amd.Index(fx.index, device=0)
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
        if "hbm_budget_bytes" in kw:
            assert not pi["sa_full"] and not pi.get("char_rank_lines"), pi
        if kw.get("rank_mode") == 1:
            assert ix.rank_mode == 1
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"]), kw
        for mo, g_noccs, g_offs in fx.locate_cases():
            noccs, offs = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(noccs, g_noccs) and np.array_equal(offs, g_offs), (kw, mo)
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

"""`-m gpu`: several handles, devices-by-name and processes sharing THIS box's one GPU: multi-device handles, striped and
range-split indexes (every "peer" on device 0), the RCCL gather with one rank, bench.py's N > 1 control flow through gloo.
tests/test_gpu_multidevice.py repeats the essentials on real peers when the box has them."""
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


@pytest.mark.parametrize("striped", [False, True])
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc"])
def test_device_chain_through_replicas_and_views(fixtures, gpu_ok, name, striped):
    """What `bench.py --gpus N` calls on every rank: femto_amd_locate_device on replica / view i of a multi-device handle
    (femto_amd_multi_child) -- with rows and in the row-free form (noccs + offsets, as parallel_locate returns them) -- for a replicated
    handle and for one whose big arrays are striped over the devices' HBM.  Goldens for every clamp."""
    fx = fixtures(name)
    g = fx.gold
    ix = femto_amd.Index(fx.index, devices=[0, 0, 0], striped=striped)
    plen, flat, starts = fx.patterns
    for i in range(3):
        v = ix.child(i)
        for mo, g_noccs, g_offs in fx.locate_cases():
            df, dl, dn, dst, do, dtot = device_locate(v, plen, flat, starts, mo, len(g_offs) + 16)
            assert dtot == len(g_offs) and np.array_equal(dn, g_noccs) and np.array_equal(do, g_offs), (i, mo)
            assert np.array_equal(df, g["count_first"]) and np.array_equal(dl, g["count_last"]), (i, mo)
            assert_row_free_equals(v, plen, flat, starts, mo, g_noccs, g_offs, (name, striped, i, mo))
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


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc", "chunks2doc"])
def test_lf_steps_device_walks_to_every_offset(fixtures, gpu_ok, name, mode):
    """femto_amd_lf_steps_device (one step of do_back_query per row, the unit the walker exchange of SURVEY.md 8(e) moves): stepping
    every row of the index until its walk ends -- offset + steps at a marked row -- gives SA[row] for every row
    (parallel_locate_range's answer), and a first step's (offset | next row) agrees with the reference's leaf goldens: a row femto
    marks reports femto's offset; an unmarked row's next row is C + Occ(L[row], row) - 1 unless L[row] is a stop character."""
    import torch
    fx = fixtures(name)
    g = fx.gold
    ix = _open(fx.index, mode)
    n = ix.info.total_length
    dev = "cuda:0"
    rows = torch.arange(n, dtype=torch.int64, device=dev)
    nxt, off = torch.empty_like(rows), torch.empty_like(rows)
    ix.lf_steps_device(n, rows.data_ptr(), nxt.data_ptr(), off.data_ptr())
    torch.cuda.synchronize()
    n0, o0 = nxt.cpu().numpy(), off.cpu().numpy()
    marked = g["off"] >= 0
    assert np.array_equal(o0[marked], g["off"][marked])                       # femto's marks are marks of every mode's
    if mode in (0, 1):
        assert (o0[~marked] == -1).all()                                       # ... and the only ones on femto's own tables
    want_sa = ix.locate_range(0, n - 1)
    assert np.array_equal(o0[o0 >= 0], want_sa[o0 >= 0])
    # the whole walk, all rows at once
    cur, steps, res = rows.clone(), torch.zeros_like(rows), torch.full_like(rows, -1)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    for _ in range(4 * int(ix.info.mark_period) + 16):
        idx = torch.nonzero(alive).flatten()
        if idx.numel() == 0:
            break
        r = cur[idx].contiguous()
        a, b = torch.empty_like(r), torch.empty_like(r)
        ix.lf_steps_device(r.numel(), r.data_ptr(), a.data_ptr(), b.data_ptr())
        torch.cuda.synchronize()
        done = b >= 0
        res[idx[done]] = b[done] + steps[idx[done]]
        dead = (~done) & (a < 0)
        alive[idx[done | dead]] = False
        go = idx[~(done | dead)]
        cur[go] = a[~(done | dead)]
        steps[go] += 1
    assert not bool(alive.any())
    assert np.array_equal(res.cpu().numpy(), want_sa)
    ix.close()


def test_walker_exchange_one_rank_on_the_gpu(fixtures, gpu_ok):
    """femto_amd/parallel.py exchange_locate with femto_amd_lf_steps_device as its step, one rank (every row is owned, the records of
    every round pass through all_to_all_single): the offsets of parallel_locate_range"""
    import torch
    import torch.distributed as dist
    from femto_amd import parallel as par
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=0, options=dict(dense_arrays=0))
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(29000 + os.getpid() % 2000)
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        n = ix.info.total_length
        rows = torch.arange(0, n, 3, dtype=torch.int64)

        def lf_step(r):
            d = r.to("cuda:0")
            a, b = torch.empty_like(d), torch.empty_like(d)
            ix.lf_steps_device(d.numel(), d.data_ptr(), a.data_ptr(), b.data_ptr())
            torch.cuda.synchronize()
            return a.cpu(), b.cpu()
        stats = {}
        got = par.exchange_locate(lf_step, rows, ix.info.block_size, ix.info.number_of_blocks, stats=stats)
        assert np.array_equal(got.numpy(), ix.locate_range(0, n - 1)[::3]) and 1 <= stats["rounds"] <= ix.info.mark_period + 3
    finally:
        if own:
            dist.destroy_process_group()
        ix.close()

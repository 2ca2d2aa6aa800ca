"""N > 1 path on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous.  Each rank runs its shard
(here through the oracle, which is test infrastructure; on the GPU box the same code path runs the
HIP kernels) and rank 0 must end up with exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, index_path, npz, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from femto_amd import parallel as par
    from oracle import pyoracle as po
    g = np.load(npz)
    plen = g["pat_len"].astype(np.int32)
    flat = g["pat_flat"].astype(np.uint16)
    starts = np.zeros(len(plen), dtype=np.int64)
    starts[1:] = np.cumsum(plen[:-1])
    o = po.Oracle(index_path)
    res = par.sharded_count(lambda a, b, c: o.count_flat(a, b, c), plen, flat, starts)
    loc = par.sharded_locate(lambda a, b, c, m: o.locate_flat(a, b, c, m), plen, flat, starts, 7)
    if rank == 0:
        ok = (np.array_equal(res[0], g["count_first"]) and np.array_equal(res[1], g["count_last"])
              and np.array_equal(loc[0], g["loc7_noccs"]) and np.array_equal(loc[1], g["loc7_offs"]))
        q.put(bool(ok))
    else:
        assert res is None and loc is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])      # 3: shards of unequal size
def test_sharded_count_and_locate_world2_gloo(fixtures, world):
    fx = fixtures("eng2doc")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fx.index, os.path.join(GOLDEN, "eng2doc.npz"), q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_range_partitions():
    from femto_amd.parallel import shard_range
    for n in (0, 1, 7, 10_000_001):
        for w in (1, 2, 3, 8):
            cover = [shard_range(n, r, w) for r in range(w)]
            assert cover[0][0] == 0 and cover[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def _wire_worker(rank, world, port, q):
    """bench.py's per-step gather: ranges narrowed to int32 (index below 2^31 rows), double buffered, async"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    n = 1000
    plen = np.full(n, 4, dtype=np.int32)
    flat = np.full(4 * n, 70, dtype=np.uint16)
    b = bench.Batch(torch, torch.device("cpu"), plen, flat)
    rng = np.random.Generator(np.random.PCG64(rank))
    first = rng.integers(0, 1 << 30, n)
    last = first + rng.integers(-1, 50, n)
    first[0], last[0] = 0, -1                      # the "no match / error" encoding must survive the narrowing
    last[5], last[77] = first[5] + 254, first[77] + 123456789      # 255 matches and more: the (pattern, count) list
    b.d_res = b.d_res2[1]
    b.d_res.copy_(torch.from_numpy(np.stack([first, last])))
    small = b.wire((1 << 30) + 1, 1)
    assert small.dtype == torch.int32 and torch.equal(small.to(torch.int64), b.d_res)
    big = b.wire((1 << 33) + 1, 1)
    assert big.dtype == torch.int64 and big is b.d_res
    out = [torch.empty_like(small) for _ in range(world)] if rank == 0 else None
    w = dist.gather(small, out, dst=0, async_op=True)
    w.wait()
    # the default payload: match counts + located offsets in one buffer (wire_results), same on every rank in size
    cap = 64
    b.offsets = torch.arange(100 * rank, 100 * rank + cap, dtype=torch.int64)
    b.d_total = torch.tensor([40 + rank, 0], dtype=torch.int64)
    bigcap = 8
    res = b.wire_results((1 << 30) + 1, cap, 1, bigcap)
    assert res.dtype == torch.uint8 and res.numel() == 16 + 16 * bigcap + 4 * cap + n      # one byte per match count
    out2 = [torch.empty_like(res) for _ in range(world)] if rank == 0 else None
    dist.gather(res, out2, dst=0, async_op=True).wait()
    if rank == 0:
        ok = True
        for r in range(world):
            g = np.random.Generator(np.random.PCG64(r))
            f = g.integers(0, 1 << 30, n)
            l_ = f + g.integers(-1, 50, n)
            f[0], l_[0] = 0, -1
            l_[5], l_[77] = f[5] + 254, f[77] + 123456789
            ok = ok and np.array_equal(out[r][0].numpy(), f) and np.array_equal(out[r][1].numpy(), l_)
            tot, cnt, off = b.unwire_results(out2[r].numpy(), (1 << 30) + 1, cap, bigcap)
            ok = ok and tot == 40 + r and np.array_equal(cnt, np.maximum(l_ - f + 1, 0)) and cnt[5] == 255 and cnt[77] == 123456790
            ok = ok and np.array_equal(off, np.arange(100 * r, 100 * r + min(cap, 40 + r)))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_wire_format_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _exchange_worker(rank, world, port, index_path, q):
    """range-split locate by WALKER EXCHANGE (femto_amd/parallel.py exchange_locate) on gloo ranks: every rank
    steps only rows it owns (the oracle's do_back_query step per row; on the GPU box femto_amd_lf_steps_device), walkers and
    results travel in one all-to-all per round"""
    import torch
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from femto_amd import parallel as par
    from oracle import pyoracle as po
    o = po.Oracle(index_path)
    n, bs, nb = o.total_length, o.block_size, o.num_blocks
    bounds = par.split_bounds(nb, world)
    stepped = {"rows": 0}

    def lf_step(rows):
        nxt, off = torch.empty_like(rows), torch.empty_like(rows)
        for i, r in enumerate(rows.tolist()):
            assert bounds[rank] <= r // bs < bounds[rank + 1], "stepped a row this rank does not own"
            ch, nr, of = o.back_step(r)
            off[i] = of
            nxt[i] = -1 if (of >= 0 or ch <= 2) else nr
        stepped["rows"] += len(rows)
        return nxt, off

    # this rank wants a strided third of ALL rows located (most of them owned by other ranks)
    want = torch.arange(rank, n, world * 3, dtype=torch.int64)
    stats = {}
    got = par.exchange_locate(lf_step, want, bs, nb, stats=stats)
    # expected: the plain walk on one process
    exp = []
    for r in want.tolist():
        steps, res = 0, -1
        while True:
            ch, nr, of = o.back_step(r)
            if of >= 0:
                res = of + steps
                break
            if ch <= 2 or steps > 4 * o.mark_period + 8:
                break
            r = nr
            steps += 1
        exp.append(res)
    ok = bool(torch.equal(got, torch.tensor(exp, dtype=torch.int64))) and stats["rounds"] <= o.mark_period + 3 and stats["records_sent"] >= len(want)
    allok = [None] * world
    dist.all_gather_object(allok, (ok, stats, stepped["rows"]))
    if rank == 0:
        q.put(allok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("acgt48k", 2), ("runs3doc", 3), ("counter400_small", 2)])
def test_walker_exchange_locate_gloo(fixtures, name, world):
    fx = fixtures(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, fx.index, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(r[0] for r in res), res
    assert all(r[2] > 0 for r in res)            # every rank stepped rows (the walkers really travelled)


def test_owner_of_rows_matches_block_ranges():
    """owner of a row = the part whose block range holds row / block_size (src/main/index.c:1613-1617; femto_amd_open_split's bounds)"""
    import torch
    from femto_amd import parallel as par
    for nb, world, bs in ((65, 8, 1 << 27), (9, 8, 1 << 27), (3, 3, 8192), (4, 2, 16384), (1, 2, 100)):
        b = par.split_bounds(nb, world)
        rows = torch.arange(0, nb * bs, max(1, bs // 3), dtype=torch.int64)
        own = par.owner_of_rows(rows, bs, b)
        for r, p in zip(rows.tolist(), own.tolist()):
            assert b[p] <= r // bs < b[p + 1], (nb, world, r, p, b)

"""Multi-GPU query sharding: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  Queries are independent (the reference's
do_parallel_query just fans out, src/main/server.c:3969-4001), so the index is replicated, the
pattern batch is split contiguously, and the ONLY collective on the path is the final gather of
results to rank 0 (16 bytes per pattern for count; sizes then payload for locate)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items owned by `rank`."""
    lo = n * rank // world
    hi = n * (rank + 1) // world
    return lo, hi


def gather_fixed(t, dst=0):
    """Gather equal-shape tensors to `dst`; returns the list on dst, None elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst)
    return out


def gather_varlen(t, dst=0):
    """Gather 1-D tensors of different lengths to `dst` (two phases: sizes, then payload padded to
    the maximum so that a single gather moves it); returns the concatenation on dst."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world == 1:
        return t
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def _shard_symbols(plen, starts, lo, hi):
    """Symbol range [f0, f1) of the flat buffer that patterns lo..hi-1 use, and their starts rebased onto it.  The C API
    accepts arbitrary (unordered, overlapping) starts, so the range is the min / max over the shard, not its ends."""
    if hi <= lo:
        return 0, 0, starts[lo:hi]
    st = np.asarray(starts[lo:hi], dtype=np.int64)
    f0 = int(st.min())
    f1 = int((st + np.asarray(plen[lo:hi], dtype=np.int64)).max())
    return f0, f1, st - f0


def sharded_count(count_fn, plen, flat, starts, device="cpu", dst=0):
    """Run count_fn(plen, flat, starts) -> (first, last) numpy int64 on this rank's contiguous shard
    of the batch and gather (first, last) for the WHOLE batch on rank dst."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(plen)
    lo, hi = shard_range(n, rank, world)
    f0, f1, s = _shard_symbols(plen, starts, lo, hi)
    first, last = count_fn(plen[lo:hi], flat[f0:f1] if f1 > f0 else flat[:0], s)
    res = torch.from_numpy(np.stack([first, last]).reshape(-1)).to(device)
    if world == 1:
        return first, last
    allres = gather_varlen(res, dst=dst)
    if rank != dst:
        return None
    out_f, out_l = [], []
    pos = 0
    allres = allres.cpu().numpy()
    for r in range(world):
        a, b = shard_range(n, r, world)
        k = b - a
        out_f.append(allres[pos:pos + k])
        out_l.append(allres[pos + k:pos + 2 * k])
        pos += 2 * k
    return np.concatenate(out_f), np.concatenate(out_l)


def sharded_locate(locate_fn, plen, flat, starts, max_occs, device="cpu", dst=0):
    """locate_fn(plen, flat, starts, max_occs) -> (noccs int32, offsets int64); gathers both."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(plen)
    lo, hi = shard_range(n, rank, world)
    f0, f1, s = _shard_symbols(plen, starts, lo, hi)
    noccs, offs = locate_fn(plen[lo:hi], flat[f0:f1] if f1 > f0 else flat[:0], s, max_occs)
    if world == 1:
        return noccs, offs
    gn = gather_varlen(torch.from_numpy(noccs.astype(np.int64)).to(device), dst=dst)
    go = gather_varlen(torch.from_numpy(offs.astype(np.int64)).to(device), dst=dst)
    if rank != dst:
        return None
    return gn.cpu().numpy().astype(np.int32), go.cpu().numpy()


def open_range_split(path, device, group=None):
    """Open `path` RANGE-SPLIT over the ranks of `group` (one process per GPU): every rank keeps the
    segment lines and block images of its own range of data blocks in HBM and maps the other ranks'
    slices (hipIpc handles exchanged with one all_gather_object) -- remote lines are then read by
    the kernels directly over xGMI; no collective runs during a query (femto_amd.h, "range-split")."""
    import femto_amd
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ix = femto_amd.Index(path, device=device, part=rank, nparts=world)
    blobs = [None] * world
    dist.all_gather_object(blobs, ix.split_export(), group=group)
    for p, blob in enumerate(blobs):
        if p != rank:
            ix.split_attach(p, blob)
    ix.split_commit()
    dist.barrier(group=group)   # nobody queries before every owner's slices are mapped everywhere
    return ix


def open_striped_shared(path, device, socket_path, group=None, devices=None):
    """Open `path` STRIPED over the GPUs of the ranks of `group` (one process per GPU): rank 0 derives the index once,
    every big array one address range whose pages are spread over all the GPUs (femto_amd_open_multi_striped), and hands
    the stripes to the other ranks as file descriptors over the Unix socket `socket_path`; every rank gets a single-GPU
    handle on its own device with all kernel families and fast paths -- remote lines travel over xGMI, no collective runs
    during a query.  `devices`: the GPU of every rank, in rank order (default: rank r uses GPU r).
    Returns (index, keepalive): keep `keepalive` referenced as long as the index is used (rank 0: the multi handle)."""
    import femto_amd
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if devices is None:
        devices = list(range(world))
    if rank == 0:
        multi = femto_amd.Index(path, devices=devices, striped=True)
        multi.striped_serve(socket_path, world - 1)
        ix = multi.child(0)
        keep = multi
    else:
        ix = femto_amd.Index(path, device=device, striped_socket=socket_path)
        keep = None
    dist.barrier(group=group)
    return ix, keep



# ---- range-split locate by WALKER EXCHANGE -----------------------------------------------------------------
# The other way to serve an index that is split over the GPUs by block range: instead of loading remote lines over xGMI
# (open_range_split / open_striped_shared above), every GPU only ever touches the rows it OWNS (owner of a row =
# row / block_size -> part, exactly bsearch_block_rows, src/main/index.c:1613-1617) and the locate walk travels: each round a
# GPU advances its resident walkers by one LF step (femto_amd_lf_steps_device), buckets them by the owner of their next row and
# all ranks exchange the buckets -- (query id, row, steps) records of 24 bytes, one all-to-all per round, at most mark distance
# + 2 rounds.  A walker that reaches a marked row travels home as a result record in the same exchange.  PROTOTYPE: which of
# the two wins on xGMI is decided by the first multi-GPU run (tools/first_multigpu.sh), not here; nothing below has run on
# more than one physical GPU.

def split_bounds(nblocks, world):
    """block boundaries of the parts, as femto_amd_open_split draws them: part p owns blocks [b[p], b[p + 1])"""
    return [nblocks * p // world for p in range(world + 1)]


def owner_of_rows(rows, block_size, bounds):
    """part owning each row: row / block_size -> the part whose block range holds it (src/main/index.c:1613-1617)"""
    blk = torch.div(rows, block_size, rounding_mode="floor")
    inner = torch.tensor(bounds[1:-1], dtype=torch.int64, device=rows.device)
    return torch.bucketize(blk, inner, right=True)


def _all_to_all_records(rec, dest, world, group=None):
    """rec: int64 [n, 3]; dest: int64 [n] in 0 .. world-1.  Every rank receives the records addressed to it (order: by sender)."""
    order = torch.argsort(dest, stable=True)
    rec = rec[order].contiguous()
    counts = torch.bincount(dest, minlength=world).to(torch.int64)
    got = torch.empty_like(counts)
    dist.all_to_all_single(got, counts, group=group)
    send_split = [int(c) for c in counts.cpu()]
    recv_split = [int(c) for c in got.cpu()]
    out = torch.empty((sum(recv_split), 3), dtype=torch.int64, device=rec.device)
    dist.all_to_all_single(out, rec, output_split_sizes=recv_split, input_split_sizes=send_split, group=group)
    return out, int(sum(send_split)) * 24


def exchange_locate(lf_step, rows, block_size, nblocks, group=None, max_rounds=4096, stats=None):
    """Text offsets of `rows` (int64 tensor: the rows THIS rank wants located, any rows of the index) by walker exchange.
    lf_step(rows_tensor) -> (next_rows, offsets): one step of the locate walk per row on rows this rank OWNS (offset >= 0: the
    row is marked and its walk ends; next < 0 and offset < 0: the walk cannot go on).  Returns an int64 tensor like `rows`
    (-1 where a walk could not finish).  stats (a dict) receives rounds, records and bytes sent by this rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = split_bounds(nblocks, world)
    dev = rows.device
    n = rows.numel()
    result = torch.full((n,), -1, dtype=torch.int64, device=dev)
    # records: [qid = home rank << 40 | slot, row (or the result's value), steps (or -1: this is a result travelling home)]
    rec = torch.stack([(rank << 40) + torch.arange(n, dtype=torch.int64, device=dev), rows.to(torch.int64),
                       torch.zeros(n, dtype=torch.int64, device=dev)], dim=1)
    dest = owner_of_rows(rec[:, 1], block_size, bounds)
    rounds, sent_records, sent_bytes = 0, 0, 0
    while True:
        live = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
        dist.all_reduce(live, group=group)
        if int(live.item()) == 0:
            break
        if rounds >= max_rounds:
            raise RuntimeError("walker exchange did not end (an LF cycle without marks?)")
        sent_records += rec.shape[0]
        got, nbytes = _all_to_all_records(rec, dest, world, group)
        sent_bytes += nbytes
        rounds += 1
        home = got[:, 2] < 0                                   # results that came home
        if bool(home.any()):
            r = got[home]
            assert bool(((r[:, 0] >> 40) == rank).all())
            result[r[:, 0] & ((1 << 40) - 1)] = r[:, 1]
        w = got[~home]
        if w.shape[0] == 0:
            rec, dest = w, torch.zeros(0, dtype=torch.int64, device=dev)
            continue
        assert bool((owner_of_rows(w[:, 1], block_size, bounds) == rank).all()), "a walker arrived at a rank that does not own its row"
        nxt, off = lf_step(w[:, 1].contiguous())
        done = off >= 0
        dead = (~done) & (nxt < 0)
        fin = done | dead
        res = torch.stack([w[fin, 0], torch.where(done[fin], off[fin] + w[fin, 2], torch.full_like(off[fin], -1)),
                           torch.full((int(fin.sum()),), -1, dtype=torch.int64, device=dev)], dim=1)
        go = torch.stack([w[~fin, 0], nxt[~fin], w[~fin, 2] + 1], dim=1)
        rec = torch.cat([res, go], dim=0)
        dest = torch.cat([res[:, 0] >> 40, owner_of_rows(go[:, 1], block_size, bounds)])
    if stats is not None:
        stats.update({"rounds": rounds, "records_sent": sent_records, "bytes_sent": sent_bytes, "world": world})
    return result

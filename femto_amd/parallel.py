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


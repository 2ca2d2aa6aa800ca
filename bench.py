#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native FM-index query engine.

Metric (BASELINE.json): patterns/sec (count+locate) on a 1 GiB-text femto index, with the rank
kernel's achieved HBM GB/s against the 8 TB/s roofline and the reference CPU path timed beside it.

Default workload = BASELINE.json configs[1]: T_acgt(2^30, seed) indexed with the reference's default
parameters (bucket 2^20 rows, block 2^27 rows, mark period 20) by this repo's own builder (GPU suffix
sort + byte-identical femto block writer); 10 M uniform random 20-mers per GPU, resident in HBM
before the timed region.  One "step" = one pass of the hot path over the batch: backward search of
every pattern (count) followed by the locate walk of every matching row (max_occs 100) -- for
random 20-mers on 1 GiB only ~0.1 % of the patterns occur, so the step is count-dominated, exactly
as configs[1] describes.  `--workload acgt_hit` / `eng` run the locate-heavy configurations.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, index replicated, each rank owns its own 10 M-pattern shard (weak
scaling); the only collective is the RCCL gather of the results to rank 0 inside every step.

The timed steps ROTATE through --rotate (default 4) distinct seeded batches per GPU, all resident before the timed region, so
that no step finds the lines of the step before it in the Infinity Cache; the same-batch replay is printed beside it once
(config.replay_same_batch).  Helpers live in benchlib/ (roofline + PMC, batch, extras, multi-GPU, CPU baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool's hosts: the driver only supports dmabuf IPC (RCCL between ranks, stripes shared between processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from benchlib import extras as bx  # noqa: E402
from benchlib.batch import Batch, _EventWork  # noqa: E402
from benchlib.common import cpu_quota, log  # noqa: E402
from benchlib.cpu_baseline import cpu_baseline  # noqa: E402
from benchlib.multigpu import multi_gpu_extras  # noqa: E402
from benchlib.roofline import add_traffic, committed_traffic, pmc_traffic, reference_format_block, roofline_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (200 x 0.54 ms: a timed region of ~0.1 s; a 20-step window is two orders below scheduler noise)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--text-log2", type=int, default=30, help="text size = 2^k bytes (30 = BASELINE configs[1])")
    ap.add_argument("--npats", type=int, default=10_000_000, help="patterns per GPU per step")
    ap.add_argument("--plen", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--len-range", default="", help="experiments: 'min,max' pattern lengths of a sampled workload instead of its own")
    ap.add_argument("--max-occs", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="patterns timed on the host CPU (0 = skip)")
    ap.add_argument("--workload", default="acgt", choices=["acgt", "acgt_hit", "eng"],
                    help="acgt = configs[1] (default, headline): random 20-mers; acgt_hit = same index, 20-mers sampled "
                         "from the text (every pattern is located); eng = configs[2]: sigma~96 text, sampled lengths 8..64")
    ap.add_argument("--layout", default="replicated", choices=["replicated", "striped", "split"],
                    help="striped: ONE index spread over the HBM of the N GPUs (BASELINE configs[4]) -- rank 0 derives it, every "
                         "big array one address range with 1/N of its pages per GPU, the other ranks map the stripes "
                         "(femto_amd.parallel.open_striped_shared); all fast paths, remote lines over xGMI.  split: the round-1 "
                         "form, 1/N of the blocks per rank through hipIpc handles, wavelet-path kernels (open_range_split)")
    ap.add_argument("--ref-sample", type=int, default=100_000, help="patterns per timed pass of the genuine reference (3 passes + warm-up)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"], help="auto: roofline.traffic from live rocprofv3 --pmc passes (N=1)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: the short run the PMC passes profile")
    ap.add_argument("--row-free", action="store_true", help="experiments / profile rounds (never the headline): the timed steps use the row-free form of "
                                                             "femto_amd_locate_device (noccs + offsets as parallel_locate returns them, no row arrays)")
    ap.add_argument("--gather", default="torch", choices=["torch", "native"],
                    help="N > 1: torch = torch.distributed.gather (RCCL); native = the library's own grouped ncclSend/ncclRecv "
                         "(femto_amd_comm_gather), its id broadcast through torch.distributed")
    ap.add_argument("--results", default="counts", choices=["counts", "ranges"],
                    help="N > 1: what is gathered to rank 0 every step: counts = match count of every pattern + the located offsets "
                         "(north_star's 'results'), ranges = the (first,last) row ranges of parallel_count")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary P_hit line (N=1 default workload only)")
    ap.add_argument("--rotate", type=int, default=4, help="distinct seeded pattern batches per GPU the timed steps rotate through (1: replay one batch)")
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "latest_pmc.json"))
    ap.add_argument("--open-opts", default="", help="experiments: femto_amd_options_t fields of the headline handle, 'name=value,...' "
                                                    "(e.g. hbm_budget_bytes=4294967296,rank_units=0)")
    args = ap.parse_args()
    open_opts = {k: int(v) for k, v in (kv.split("=") for kv in args.open_opts.split(",") if kv)}
    # the benchmark's handles take whatever HBM is free (dense suffix arrays, deepest tables: rounds 1-4's rule and numbers); the
    # library's DEFAULT is a bound (include/femto_amd.h hbm_budget_bytes) -- measured by the `default_open` extra
    open_opts.setdefault("hbm_budget_bytes", -2)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # FEMTO_AMD_BENCH_BACKEND=gloo: control-flow smoke test of the N > 1 path on a box with fewer GPUs than ranks
    # (ranks share devices, the gather travels through host memory); never a measurement.
    backend = os.environ.get("FEMTO_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import femto_amd
    from femto_amd import textgen as tg

    eng = args.workload == "eng"
    hit = args.workload != "acgt"
    want_extra = (args.workload == "acgt" and world == 1 and not args.no_extra and not args.pmc_child)
    need_text = hit or want_extra
    n_text = 1 << args.text_log2
    os.makedirs(args.workdir, exist_ok=True)
    index_path = os.path.join(args.workdir, f"{'eng' if eng else 'acgt'}_2p{args.text_log2}_s{args.seed}")
    text_path = index_path + ".text.npy"
    build_s = 0.0
    if rank == 0:
        t0 = time.time()
        have_index = os.path.exists(os.path.join(index_path, "_femto_index"))
        have_text = os.path.exists(text_path)
        if not have_index or (need_text and not have_text):
            text = tg.t_eng_torch(n_text, args.seed, f"cuda:{local_rank}") if eng else tg.t_acgt(n_text, args.seed)
            t1 = time.time()
            if need_text and not have_text:
                np.save(text_path, text)
            if not have_index:
                t2 = time.time()
                femto_amd.build_index(index_path, [text], params=None, infos=["bench"], device=local_rank)
                log(f"text {t1 - t0:.1f}s, index build {time.time() - t2:.1f}s -> {index_path}")
            del text
        build_s = time.time() - t0
    if world > 1:
        dist.barrier()
    if world > 1 and args.layout in ("split", "striped") and backend == "nccl":
        # these layouts load lines that live in other GPUs' HBM: every pair must be able to (say so now, readably)
        ndev_ = torch.cuda.device_count()
        bad = [(local_rank, j) for j in range(min(world, ndev_)) if j != local_rank and not torch.cuda.can_device_access_peer(local_rank, j)]
        if bad:
            raise RuntimeError(f"--layout {args.layout}: no peer access between GPU pairs {bad} (hipDeviceCanAccessPeer); "
                               "use --layout replicated on this node")
    t0 = time.time()
    ix_keep = None
    if args.layout == "split" and world > 1:
        from femto_amd import parallel as fpar
        ix = fpar.open_range_split(index_path, local_rank)
    elif args.layout == "striped" and world > 1:
        from femto_amd import parallel as fpar
        ndev = torch.cuda.device_count()
        ix, ix_keep = fpar.open_striped_shared(index_path, local_rank, os.path.join(args.workdir, "stripes.sock"),
                                               devices=[r % ndev for r in range(world)])
    else:
        ix = femto_amd.Index(index_path, device=local_rank, options=open_opts)
    open_s = time.time() - t0
    info = ix.info

    # synthetic patterns, resident in HBM before the timed region
    npats = args.npats
    if hit:
        text = np.load(text_path, mmap_mode="r")
        kmin, kmax = (8, 64) if eng else (args.plen, args.plen)
        if args.len_range:      # experiments only: the named workloads use the lengths above
            kmin, kmax = (int(x) for x in args.len_range.split(","))
        plen, flat = tg.p_hit(kmin, kmax, npats, args.seed + 1000 + rank, np.asarray(text))
        del text
    else:
        plen, flat = tg.p_rand(args.plen, npats, args.seed + 1000 + rank)
    batch = Batch(torch, dev, plen, flat)
    nsets = 1 if args.pmc_child else max(1, args.rotate)
    for k in range(1, nsets):                # further batches of the same workload, seeds of their own (sets x ranks never collide)
        if hit:
            text = np.load(text_path, mmap_mode="r")
            p_k, f_k = tg.p_hit(kmin, kmax, npats, args.seed + 1000 + rank + 7919 * k, np.asarray(text))
            del text
        else:
            p_k, f_k = tg.p_rand(args.plen, npats, args.seed + 1000 + rank + 7919 * k)
        batch.add_inputs(p_k, f_k)
        del p_k, f_k
    direct = ix.rank_mode in (3, 4)      # the caller-order pipeline (direct_kernels.hip.hpp)
    if args.pmc_child:      # the short run the PMC passes of pmc_traffic() profile: same index, same batch, a few steps
        if os.environ.get("FEMTO_AMD_BENCH_CHILD_INFO"):
            with open(os.environ["FEMTO_AMD_BENCH_CHILD_INFO"], "w") as fh:
                json.dump(ix.pack_info(), fh)
        st = torch.cuda.current_stream().cuda_stream
        batch.row_free = args.row_free
        batch.settle(ix, args.max_occs, st)
        for _ in range(args.warmup + args.steps):
            batch.step(ix, args.max_occs, st)
        torch.cuda.synchronize()
        ix.close()
        return
    cap, bigcap = 0, 1024
    if world > 1 and args.results == "counts":     # untimed: every rank's row total, the common capacity of the gathered offsets
        batch.settle(ix, args.max_occs, torch.cuda.current_stream().cuda_stream)
        tcap = torch.tensor([batch.max_total], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tcap, op=dist.ReduceOp.MAX)
        cap = int(tcap.item() * 1.25) + 1024
        if batch.offsets.numel() < cap:
            batch.offsets = torch.empty(cap, dtype=torch.int64, device=dev)
        nbig = ((batch.d_res[1] - batch.d_res[0]) >= 254).sum().to(tcap.device).reshape(1)
        dist.all_reduce(nbig, op=dist.ReduceOp.MAX)
        bigcap = int(nbig.item() * 1.25) + 1024

    def payload_of(b=0):
        if args.results == "counts":
            return batch.wire_results(info.total_length, cap, b, bigcap, ix, torch.cuda.current_stream().cuda_stream)
        return batch.wire(info.total_length, b)

    gather_lists = None
    if world > 1 and rank == 0:
        gather_lists = [[torch.empty_like(payload_of(), device=None if backend == "nccl" else "cpu")
                         for _ in range(world)] for _ in range(2)]
    native = world > 1 and args.gather == "native" and backend == "nccl"
    gstream, recv_native = None, None
    if native:      # the C ABI's own gather: grouped point-to-point transfers on a stream of its own
        ids = [femto_amd.Index.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ix.comm_init(ids[0], world, rank)
        gstream = torch.cuda.Stream()
        w0 = payload_of()
        if rank == 0:
            recv_native = [torch.empty((world,) + tuple(w0.shape), dtype=w0.dtype, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    pending = [None, None]
    counter = {"k": 0}
    stalls = []         # (event before, event after) every wait for a gather: how long the compute stream stood still

    def step():
        # Results are double buffered: the RCCL gather of step k (over xGMI, on RCCL's own stream, ordered
        # after the kernels of step k) overlaps the search kernels of step k+1, which write the other buffer.
        b = counter["k"] & 1
        batch.use(counter["k"])         # the next of the resident batches
        counter["k"] += 1
        if pending[b] is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pending[b].wait()          # the buffer's previous gather must be done before it is overwritten
            e1.record()
            stalls.append((e0, e1))
            pending[b] = None
        batch.step(ix, args.max_occs, stream, b)
        if native:
            payload = payload_of(b)
            ev = torch.cuda.Event()
            ev.record()
            gstream.wait_event(ev)
            ix.comm_gather(payload.data_ptr(), recv_native[b].data_ptr() if rank == 0 else 0, payload.numel() * payload.element_size(), 0,
                           gstream.cuda_stream)
            done = torch.cuda.Event()
            done.record(gstream)
            pending[b] = _EventWork(done)
        elif world > 1:
            payload = payload_of(b)
            if backend != "nccl":
                payload = payload.cpu()
            pending[b] = dist.gather(payload, gather_lists[b] if rank == 0 else None, dst=0, async_op=True)

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    batch.settle(ix, args.max_occs, stream)      # untimed: sizes the offsets buffer (the timed steps never read the total back)
    batch.row_free = args.row_free
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ix.kernel_time_reset()
    ix.kernel_time_enable(True)
    torch.cuda.synchronize()
    stalls.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                       # every step's gather has landed on rank 0 inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ix.kernel_time_enable(False)
    if args.row_free:      # everything below reads (first, last) of the last step: once more with rows, untimed
        batch.row_free = False
        batch.use(counter["k"] - 1)
        batch.step(ix, args.max_occs, stream, (counter["k"] - 1) & 1)
        torch.cuda.synchronize()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    cnt_ms, cnt_n = ix.kernel_time("count")
    loc_ms, loc_n = ix.kernel_time("locate")

    # ---- N > 1: what every rank saw, so that ONE run explains an efficiency below 1 (search kernels vs waiting for the gather)
    per_rank, multi_extra = None, None
    if world > 1:
        own_elapsed = time.perf_counter() - t0      # (includes the barrier: the spread between ranks shows who was waited for)
        stall_ms = sum(a.elapsed_time(b) for a, b in stalls) if stalls and backend == "nccl" else 0.0
        pl = payload_of((counter["k"] - 1) & 1) if counter["k"] else None
        mine = {"rank": rank, "device": local_rank, "count_kernel_ms": cnt_ms, "locate_kernel_ms": loc_ms,
                "search_ms_per_step": cnt_ms + loc_ms, "gather_stall_ms_per_step": stall_ms / max(1, args.steps),
                "gather_waits": len(stalls), "gather_payload_bytes": int(pl.numel() * pl.element_size()) if pl is not None else 0,
                "located_rows": int(batch.d_total[0].item()), "own_wall_s": own_elapsed, "world_size_seen": dist.get_world_size(),
                "native_comm": ix.comm_info() if native else None}
        allr = [None] * world if rank == 0 else None
        dist.gather_object(mine, allr, dst=0)
        per_rank = allr
        multi_extra = multi_gpu_extras(args, torch, dist, femto_amd, ix, batch, rank, world, local_rank, dev, backend, native, payload_of,
                                       index_path, elapsed, npats, per_rank)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    batch.total = int(batch.d_total[0].item())      # the LAST step's row total (the steps rotate through batches of different totals)
    first = batch.d_res[0].cpu().numpy()
    last = batch.d_res[1].cpu().numpy()
    g_noccs = batch.d_noccs.cpu().numpy()
    g_ost = batch.d_ostarts.cpu().numpy()
    g_offs = batch.offsets[:batch.total].cpu().numpy()
    located_rows = batch.total
    value = world * npats * args.steps / elapsed
    gathered_ok = None
    if world > 1 and args.results == "counts" and counter["k"]:
        # What arrived on rank 0 in the last step, checked slot by slot: rank 0 regenerates EVERY rank's shard of that step from
        # its seed, searches it on its own GPU and compares the whole slot -- the row total, the match count of every pattern
        # and every located offset (a transposed, stale or half-written peer buffer fails here; round-5 verdict, task 1b).
        b_last = (counter["k"] - 1) & 1
        k_last = (counter["k"] - 1) % nsets
        slots = recv_native[b_last] if native else gather_lists[b_last]
        gathered_ok = True
        checked = []
        for r in range(world):
            tot_r, cnt_r, off_r = batch.unwire_results(slots[r].cpu().numpy(), info.total_length, cap, bigcap)
            if r == 0:
                gathered_ok = (gathered_ok and tot_r == batch.total and np.array_equal(cnt_r, np.maximum(last - first + 1, 0))
                               and np.array_equal(off_r, g_offs[:len(off_r)]))
            seed_r = args.seed + 1000 + r + 7919 * k_last
            if hit:
                text = np.load(text_path, mmap_mode="r")
                p_r, f_r = tg.p_hit(kmin, kmax, npats, seed_r, np.asarray(text))
                del text
            else:
                p_r, f_r = tg.p_rand(args.plen, npats, seed_r)
            vb = Batch(torch, dev, p_r, f_r)
            vb.settle(ix, args.max_occs, stream)
            torch.cuda.synchronize()
            v_cnt = (vb.d_res[1] - vb.d_res[0] + 1).clamp_(min=0).cpu().numpy()
            v_off = vb.offsets[:vb.total].cpu().numpy()
            same = tot_r == vb.total and np.array_equal(cnt_r, v_cnt) and np.array_equal(off_r, v_off[:len(off_r)]) and len(off_r) == min(vb.total, cap)
            checked.append({"rank": r, "patterns": int(npats), "rows": int(vb.total), "equal": bool(same)})
            gathered_ok = gathered_ok and same
            del vb, p_r, f_r
        log("gathered results verified against a local search of every rank's shard:", checked)
        assert gathered_ok, f"the gathered results differ from a local search of the ranks' shards: {checked}"

    # ---- everything below checks and re-measures input set 0 (the patterns `plen` / `flat` of this process)
    replay = None
    if nsets > 1:
        batch.use(0)
        batch.step(ix, args.max_occs, stream, 0)
        torch.cuda.synchronize()
        if world == 1 and not args.pmc_child:      # the same batch replayed, printed once beside the rotating value
            t0 = time.perf_counter()
            for _ in range(args.steps):
                batch.step(ix, args.max_occs, stream, 0)
            torch.cuda.synchronize()
            re_s = time.perf_counter() - t0
            replay = {"value": npats * args.steps / re_s, "ms_per_step": 1e3 * re_s / args.steps, "steps": args.steps,
                      "what": "ONE batch replayed every step (rounds 1-4 timed this); the headline value rotates through distinct batches"}
        first = batch.d_res[0].cpu().numpy()
        last = batch.d_res[1].cpu().numpy()
        g_noccs = batch.d_noccs.cpu().numpy()
        g_ost = batch.d_ostarts.cpu().numpy()
        batch.total = int(batch.d_total[0].item())
        g_offs = batch.offsets[:batch.total].cpu().numpy()
        located_rows = batch.total
    ctx = bx.Ctx(args=args, torch=torch, femto_amd=femto_amd, tg=tg, dev=dev, local_rank=local_rank, stream=stream, index_path=index_path,
                 text_path=text_path, n_text=n_text, npats=npats)
    res_ref = (first, last, g_noccs, g_ost, g_offs)

    # ---- secondary line: every pattern occurs and is located (same index, P_hit 20-mers)
    extra = None
    if want_extra:
        text = np.load(text_path, mmap_mode="r")
        hp, hf = tg.p_hit(args.plen, args.plen, npats, args.seed + 2000, np.asarray(text))
        del text
        hb = Batch(torch, dev, hp, hf)
        extra = {}

        def guarded(name, fn):
            try:
                extra[name] = fn()
            except AssertionError:
                raise                      # a parity failure is never swallowed
            except Exception as ex:      # noqa: BLE001
                extra[name] = {"error": repr(ex)}
        guarded("p_hit_count_locate", lambda: bx.p_hit_extra(ctx, ix, hb))
        hit_rows = hb.total
        del hb
        guarded("compact_keys_count_locate", lambda: bx.keys_extra(ctx, ix, batch, res_ref))
        guarded("regexp_batch", lambda: bx.regexp_extra(ctx, ix))
        guarded("host_pointer_count", lambda: bx.host_pointer_extra(ctx, ix, batch, plen, flat, first, last))
        extra.update(bx.shim_extras(args, index_path, plen, flat, first, last, located_rows))
        guarded("shim_parallel_locate_sampled", lambda: bx.shim_locate_sampled_extra(args, index_path, hp, hf, hit_rows))
        del hp, hf
    # ---- CPU baseline + bit-exact check on a bounded sample of the same batch (rank 0 only)
    cpu, ref_work, cd_count, csub = cpu_baseline(args, index_path, plen, batch.starts, flat, first, last, g_noccs, g_ost, g_offs, value)

    # ---- roofline of the dominant kernel.  achieved = COMPULSORY bytes per launch / average kernel duration, where the
    # compulsory bytes are what the launch must move at least once: 128 B for every DISTINCT line of a derived array it
    # loads (counted on the GPU by a traced run of the same batch, femto_amd_trace_lines) + the arrays it streams
    # exactly once (pattern lengths / starts / symbols in, ranges and row counts out).  frac <= 1 by construction.
    roof = None
    if cnt_n > 0:
        try:
            if args.row_free:
                ix.set_option("trace_row_free", 1)      # the traced twins follow the form the steps were timed in
            roof, kname, k_ms, comp, step_bytes = roofline_block(ix, direct, batch, npats, plen, args.max_occs, cnt_ms, loc_ms, cnt_n)
        except femto_amd.FemtoAmdError as ex:      # the line trace follows the packed modes' pipeline: modes 0 / 1 report times only
            roof = None
            log("no roofline block:", ex)
    if roof is not None:
        traffic, traffic_src = None, None
        if args.pmc != "off" and world == 1:
            try:
                traffic, traffic_src = pmc_traffic(args, kname, ix.pack_info())
            except Exception as ex:      # noqa: BLE001
                log("pmc pass failed:", repr(ex))
        if traffic is None:
            traffic, traffic_src = committed_traffic(args, kname, npats)
        add_traffic(roof, traffic, traffic_src, k_ms, comp)
        roof["whole_step_GBs"] = step_bytes / (1e-3 * 1e3 * elapsed / args.steps) / 1e9   # cross-check: compulsory bytes of the step / ms_per_step < peak
        if cd_count:
            roof["reference_format"] = reference_format_block(cd_count, csub, npats, cnt_ms)
            roof["reference_format"]["note"] = "femto's own algorithm on femto's own format; x_peak > 1: the timed kernel does not do that work (see compulsory / line_reads)"
    if want_extra:
        extra["budget4x"] = bx.budget_extra(ctx, batch, plen, res_ref)
        extra["default_open"] = bx.budget_extra(ctx, batch, plen, res_ref, budget="default")
        extra["mode1_wavelet_tree"] = bx.mode1_extra(ctx, ix, batch, res_ref, cd_count, csub)
        extra["mode0_wavefront_per_query"] = bx.mode0_extra(ctx, ix, batch, res_ref)
        if roof is not None:      # compact copies inside the block the driver keeps
            b4 = extra["budget4x"]
            br = b4.get("roofline") or {}
            roof["budget4x"] = {"value": b4.get("value"), "ms_per_step": b4.get("ms_per_step"), "kernel_ms": br.get("kernel_ms"),
                                "frac_distinct_lines": br.get("frac"), "frac_line_reads": (br.get("line_reads") or {}).get("frac"),
                                "traffic_GBs": br.get("traffic_GBs"), "hbm_held": (b4.get("structures") or {}).get("hbm_allocated"),
                                "level_table_syms": (b4.get("structures") or {}).get("level_table_syms"),
                                "p_hit_value": (b4.get("p_hit") or {}).get("value"), "error": b4.get("error")}
            d0 = extra["default_open"]
            dr = d0.get("roofline") or {}
            roof["default_open"] = {"value": d0.get("value"), "ms_per_step": d0.get("ms_per_step"), "kernel_ms": dr.get("kernel_ms"),
                                    "frac_distinct_lines": dr.get("frac"), "frac_line_reads": (dr.get("line_reads") or {}).get("frac"),
                                    "traffic_GBs": dr.get("traffic_GBs"), "hbm_held": (d0.get("structures") or {}).get("hbm_allocated"),
                                    "hbm_budget": (d0.get("structures") or {}).get("hbm_budget"),
                                    "level_table_syms": (d0.get("structures") or {}).get("level_table_syms"),
                                    "p_hit_value": (d0.get("p_hit") or {}).get("value"), "error": d0.get("error")}
            m1 = extra["mode1_wavelet_tree"]
            roof["mode1"] = {"value": m1.get("value"), "count_kernel_ms": m1.get("count_kernel_ms"), "frac_reference_format": (m1.get("roofline") or {}).get("frac"),
                             "traffic_GBs": (m1.get("roofline") or {}).get("traffic_GBs"), "error": m1.get("error")}
            m0 = extra["mode0_wavefront_per_query"]
            roof["mode0"] = {"value": m0.get("value"), "count_kernel_ms": m0.get("count_kernel_ms"), "error": m0.get("error")}

    # BASELINE configs[2] as an extra line, LAST: the headline index is closed first, so that the sigma~96 index is opened with
    # the whole HBM to budget against (opened next to the 79 GB DNA index its wide context table got the denser, slower layout)
    main_rank_mode, main_pack_info, main_structs = ix.rank_mode, ix.pack_info(), ix.structures()
    if want_extra:
        del batch
        ix.close()
        ix = None
        torch.cuda.empty_cache()
        # BASELINE configs[2] (benchlib/extras.py cfg3_extra).  A failure here must not cost the headline line -- but a parity
        # failure does: a fast step with wrong answers is not a result.
        try:
            extra["cfg3_text96_count_locate"] = bx.cfg3_extra(ctx, world)
        except AssertionError:
            raise
        except Exception as ex:      # noqa: BLE001
            extra["cfg3_text96_count_locate"] = {"error": repr(ex)}

    if roof is not None:
        # Scalar copies at the top level of the block (a reader that keeps only scalars -- the driver's record -- still sees what
        # `frac` is and is not; round-5 verdict, task 5).  DESIGN.md section 4 defines the three byte counts.
        ex_ = extra or {}

        def val(name, key="value"):
            o = ex_.get(name)
            return o.get(key) if isinstance(o, dict) else None
        roof["frac_model"] = "distinct_lines"
        roof["frac_line_reads"] = (roof.get("line_reads") or {}).get("frac")
        roof["frac_counter_traffic"] = (roof["traffic_GBs"] / roof["peak"]) if roof.get("traffic_GBs") else None
        roof["x_peak_survey_8d"] = (roof.get("reference_format") or {}).get("x_peak")
        roof["hbm_held_bytes"] = main_structs.get("hbm_allocated")
        roof["default_open_value"] = val("default_open")
        roof["default_open_hbm_held_bytes"] = (val("default_open", "structures") or {}).get("hbm_allocated")
        roof["budget4x_value"] = val("budget4x")
        roof["p_hit_value"] = val("p_hit_count_locate")
        roof["p_hit_frac"] = (val("p_hit_count_locate", "roofline") or {}).get("frac")
        roof["p_hit_row_free_value"] = (val("p_hit_count_locate", "row_free") or {}).get("value")
        roof["p_hit_row_free_frac"] = ((val("p_hit_count_locate", "row_free") or {}).get("roofline") or {}).get("frac")
        roof["cfg3_row_free_value"] = (val("cfg3_text96_count_locate", "row_free") or {}).get("value")
        roof["cfg3_row_free_frac"] = ((val("cfg3_text96_count_locate", "row_free") or {}).get("roofline") or {}).get("frac")
        roof["cfg3_default_open_row_free_value"] = ((val("cfg3_text96_count_locate", "default_open") or {}).get("row_free") or {}).get("value")
        roof["cfg3_value"] = val("cfg3_text96_count_locate")
        roof["cfg3_frac"] = (val("cfg3_text96_count_locate", "roofline") or {}).get("frac")
        roof["cfg3_default_open_value"] = (val("cfg3_text96_count_locate", "default_open") or {}).get("value")
        roof["mode1_value"] = val("mode1_wavelet_tree")
        roof["mode1_x_peak_survey_8d"] = (val("mode1_wavelet_tree", "roofline") or {}).get("frac")
        roof["mode0_value"] = val("mode0_wavefront_per_query")
    if cpu:
        cpu["cgroup_cpu_quota"] = cpu_quota()      # CPUs this container may use on average (None: no quota); `cores` above is what was used
        cpu["gpu_vs_cpu"] = value / cpu["value"]     # a baseline, not a quality measure: the roofline fraction is
    wl = {"acgt": f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_rand 20-mers per GPU, count()+locate(max_occs={args.max_occs})",
          "acgt_hit": f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_hit 20-mers per GPU, count()+locate(max_occs={args.max_occs})",
          "eng": f"T_eng(2^{args.text_log2}) femto index (default params), {npats} P_hit lengths 8..64 per GPU, count()+locate(max_occs={args.max_occs})"}[args.workload]
    out = {
        "metric": "patterns/sec (count+locate) on " + ("1 GiB index" if args.text_log2 == 30 else f"2^{args.text_log2} B index"),
        "value": value, "unit": "patterns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": wl + (" [ROW-FREE form: noccs + offsets, no rows]" if args.row_free else ""), "row_free": bool(args.row_free), "batches_rotated": nsets, "replay_same_batch": replay, "text_bytes": n_text, "patterns_per_gpu": npats, "pattern_len": args.plen, "seed": args.seed,
                   "located_rows_per_gpu": located_rows, "gathered_results_verified": gathered_ok, "per_rank": per_rank, "matched_patterns_frac": float(np.mean(last >= first)),
                   "rank_mode": {4: "pack2", 3: "pack", 1: "lane", 0: "raw"}[main_rank_mode], "index": {"rows": int(info.total_length), "blocks": int(info.number_of_blocks), "buckets": int(info.total_buckets),
                             "image_bytes": int(info.image_bytes), "table_bytes": int(info.table_bytes),
                             "packed_lines": main_pack_info, "structures": main_structs},
                   "parallelism": ("range-split index (1/N of the blocks per GPU, peer loads over xGMI)" if args.layout == "split" and world > 1 else
                                   "striped index (every big array 1/N per GPU, one address range, shared between the ranks; remote lines over xGMI)"
                                   if args.layout == "striped" and world > 1 else "replicated index") + f", query shards x{world}" + ((", RCCL gather of the results to rank 0 every step (32-bit when the index has < 2^31 rows), overlapped with the next step's kernels"
                                                                                  + ("; gather = femto_amd_comm_gather (grouped ncclSend/ncclRecv)" if native else "; gather = torch.distributed.gather") + (
                                                                                  "; payload = match counts + located offsets" if args.results == "counts" else "; payload = (first,last) ranges")) if world > 1 else ""),
                   "build_s": build_s, "open_s": open_s},
        "roofline": roof, "cpu_baseline": cpu, "reference_equivalent_work": ref_work,
    }
    # every extra measurement is a short JSON line of its own, BEFORE the headline line (which stays last and small enough
    # for the driver to keep whole); the headline names them and repeats their values
    ex_all = extra if extra is not None else (multi_extra or {})
    for name, obj in ex_all.items():
        print(json.dumps({"extra": name, **(obj if isinstance(obj, dict) else {"value": obj})}), flush=True)
    out["extras"] = {name: ({k: obj.get(k) for k in ("value", "unit", "ms_per_step", "ms", "error") if obj.get(k) is not None} if isinstance(obj, dict) else obj)
                     for name, obj in ex_all.items()}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

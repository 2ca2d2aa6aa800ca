#!/usr/bin/env python3
"""tools/exchange_bench.py -- range-split locate by WALKER EXCHANGE against direct loads, one process per GPU (torchrun).

SURVEY.md 8(e) names two ways to serve an index split over the GPUs by block range: exchange the walkers (every GPU steps only
the rows it owns; (query, row, steps) records travel in one all-to-all per LF round) or load remote lines over xGMI.  This
script times both on the same batch and checks that they return the same offsets:

  * replicated:  femto_amd_locate_device on a replicated handle (no traffic between the GPUs; the baseline)
  * exchange:    count on the local handle, then femto_amd/parallel.py exchange_locate -- every LF step on the rank that
                 OWNS the row (asserted), walkers exchanged with torch.distributed.all_to_all_single (RCCL)

PROTOTYPE: the handle of every rank still holds the whole index (ownership is enforced, memory is not saved), so that the
cost of the EXCHANGE can be read off the first multi-GPU run before anything is built around it.  With one rank it runs as a
self-test (every row is owned, nothing travels).  Prints one JSON line on rank 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/exchange_bench.py [--text-log2 30]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=30)
    ap.add_argument("--npats", type=int, default=2_000_000)
    ap.add_argument("--max-occs", type=int, default=100)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--budget", type=int, default=0, help="hbm_budget_bytes of the handles (0: the library's default bound -- marks, no suffix array: walks of several steps)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import femto_amd
    from femto_amd import parallel as par
    from femto_amd import textgen as tg
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("FEMTO_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    path = os.path.join(args.workdir, f"acgt_2p{args.text_log2}_s{args.seed}")
    tpath = path + ".text.npy"
    if rank == 0 and not (os.path.exists(os.path.join(path, "_femto_index")) and os.path.exists(tpath)):
        os.makedirs(args.workdir, exist_ok=True)
        text = tg.t_acgt(1 << args.text_log2, args.seed)
        np.save(tpath, text)
        if not os.path.exists(os.path.join(path, "_femto_index")):
            femto_amd.build_index(path, [text], params=None, infos=["bench"], device=local)
        del text
    dist.barrier()
    ix = femto_amd.Index(path, device=local, options={"hbm_budget_bytes": args.budget} if args.budget else None)
    info = ix.info
    text = np.load(tpath, mmap_mode="r")
    plen, flat = tg.p_hit(20, 20, args.npats, args.seed + 5000 + rank, np.asarray(text))
    del text
    n = len(plen)
    starts = tg.starts_of(plen)
    d_plen, d_flat, d_starts = torch.from_numpy(plen).to(dev), torch.from_numpy(flat.view(np.int16)).to(dev), torch.from_numpy(starts).to(dev)
    f, l = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
    noccs = torch.zeros(n, dtype=torch.int32, device=dev)
    ost = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    cap = n * 2 + 1024
    offs = torch.full((cap,), -7, dtype=torch.int64, device=dev)
    total = torch.zeros(2, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def replicated():
        ix.locate_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), args.max_occs, f.data_ptr(), l.data_ptr(), noccs.data_ptr(),
                         ost.data_ptr(), offs.data_ptr(), cap, total.data_ptr(), st)

    def lf_step(rows):
        nxt, off = torch.empty_like(rows), torch.empty_like(rows)
        ix.lf_steps_device(rows.numel(), rows.data_ptr(), nxt.data_ptr(), off.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return nxt, off

    stats = {}

    def exchange():
        ix.locate_plan_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), args.max_occs, f.data_ptr(), l.data_ptr(), noccs.data_ptr(),
                              ost.data_ptr(), st)
        k = noccs.to(torch.int64)
        tot = int(ost[n].item())
        rows = torch.repeat_interleave(f, k) + (torch.arange(tot, dtype=torch.int64, device=dev) - torch.repeat_interleave(ost[:n], k))
        return par.exchange_locate(lf_step, rows, info.block_size, info.number_of_blocks, stats=stats)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()) / args.steps, out

    t_rep, _ = timed(replicated)
    tot = int(total[0].item())
    want = offs[:tot].clone()
    t_exc, got = timed(exchange)
    same = bool(tot == got.numel() and torch.equal(got, want))
    allsame = [None] * world
    dist.all_gather_object(allsame, (same, tot, dict(stats)))
    if rank == 0:
        print(json.dumps({"what": "range-split locate: walker exchange (SURVEY 8(e)) vs a replicated handle, same batch", "n_gpus": world, "patterns_per_gpu": n,
                          "rows_per_gpu": [a[1] for a in allsame], "replicated_ms_per_step": 1e3 * t_rep, "exchange_ms_per_step": 1e3 * t_exc,
                          "replicated_patterns_per_s": world * n / t_rep, "exchange_patterns_per_s": world * n / t_exc,
                          "exchange_rounds": [a[2].get("rounds") for a in allsame], "exchange_bytes_sent_per_rank": [a[2].get("bytes_sent") for a in allsame],
                          "offsets_equal": [a[0] for a in allsame], "structures": ix.structures(),
                          "note": "prototype: every handle holds the whole index; ownership (row / block_size -> part) is enforced on every LF step"}), flush=True)
    assert same, "walker exchange: offsets differ from the replicated handle's"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

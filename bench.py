#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native FM-index query engine.

Metric (BASELINE.json): patterns/sec for batched count() on a 1 GiB-text femto index, with the rank
kernel's achieved HBM GB/s against the 8 TB/s roofline, and the reference CPU path timed beside it.

Default workload (BASELINE.json configs[1]): T_acgt(2^30, seed) indexed with the reference's default
parameters (bucket 2^20 rows, block 2^27 rows, mark period 20) by this repo's own builder (GPU suffix
sort + byte-identical femto block writer); 10 M uniform random 20-mers per GPU, already resident in
HBM when the timed region starts.  One "step" = one femto_amd_count_device() pass over the batch.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, index replicated, each rank owns its own 10 M-pattern shard (weak
scaling); the only collective is the RCCL gather of (first,last) to rank 0 inside every step.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def build_or_reuse_index(path, text_fn, params, device):
    import femto_amd
    marker = os.path.join(path, "_femto_index")
    if os.path.exists(marker):
        return 0.0
    t0 = time.time()
    text = text_fn()
    t1 = time.time()
    femto_amd.build_index(path, [text], params=params, infos=["bench"], device=device)
    log(f"text {t1 - t0:.1f}s, index build {time.time() - t1:.1f}s -> {path}")
    return time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--text-log2", type=int, default=30, help="text size = 2^k bytes (30 = BASELINE config 2)")
    ap.add_argument("--npats", type=int, default=10_000_000, help="patterns per GPU per step")
    ap.add_argument("--plen", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="patterns timed on the host CPU (0 = skip)")
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "latest_pmc.json"))
    ap.add_argument("--workload", default="acgt_count", choices=["acgt_count", "acgt_locate", "eng_locate"],
                    help="acgt_count = BASELINE configs[1] (default, the headline line); acgt_locate = same index, "
                         "sampled 20-mers, count+locate; eng_locate = configs[2]: sigma~96 text, lengths 8..64, count+locate")
    ap.add_argument("--max-occs", type=int, default=100)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import femto_amd
    from femto_amd import textgen as tg

    n_text = 1 << args.text_log2
    os.makedirs(args.workdir, exist_ok=True)
    eng = args.workload == "eng_locate"
    locate = args.workload != "acgt_count"
    index_path = os.path.join(args.workdir, f"{'eng' if eng else 'acgt'}_2p{args.text_log2}_s{args.seed}")
    text_path = index_path + ".text.npy"
    build_s = 0.0
    if rank == 0:
        def make_text():
            t = tg.t_eng_torch(n_text, args.seed, f"cuda:{local_rank}") if eng else tg.t_acgt(n_text, args.seed)
            if locate:
                np.save(text_path, t)
            return t
        if locate and not os.path.exists(text_path) and os.path.exists(os.path.join(index_path, "_femto_index")):
            make_text()
        build_s = build_or_reuse_index(index_path, make_text, None, local_rank)
    if world > 1:
        dist.barrier()
    t0 = time.time()
    ix = femto_amd.Index(index_path, device=local_rank)
    open_s = time.time() - t0
    info = ix.info

    # synthetic patterns, resident in HBM before the timed region
    npats = args.npats
    if locate:
        if world > 1:
            dist.barrier()
        text = np.load(text_path, mmap_mode="r")
        kmin, kmax = (8, 64) if eng else (args.plen, args.plen)
        plen, flat = tg.p_hit(kmin, kmax, npats, args.seed + 1000 + rank, np.asarray(text))
        del text
    else:
        plen, flat = tg.p_rand(args.plen, npats, args.seed + 1000 + rank)
    starts = tg.starts_of(plen)
    d_plen = torch.from_numpy(plen).to(dev)
    d_flat = torch.from_numpy(flat.view(np.int16)).to(dev)
    d_starts = torch.from_numpy(starts).to(dev)
    d_res = torch.empty((2, npats), dtype=torch.int64, device=dev)   # [first; last]
    gather_list = None
    if world > 1 and rank == 0:
        gather_list = [torch.empty_like(d_res) for _ in range(world)]
    stream = torch.cuda.current_stream().cuda_stream

    d_noccs = torch.empty(npats, dtype=torch.int32, device=dev) if locate else None
    d_ostarts = torch.empty(npats + 1, dtype=torch.int64, device=dev) if locate else None
    loc = {"offsets": None, "total": 0}

    def step():
        if not locate:
            ix.count_device(npats, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(),
                            d_res[0].data_ptr(), d_res[1].data_ptr(), stream)
        else:
            ix.locate_plan_device(npats, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), args.max_occs,
                                  d_res[0].data_ptr(), d_res[1].data_ptr(), d_noccs.data_ptr(), d_ostarts.data_ptr(), stream)
            total = int(d_ostarts[npats].item())      # sizes the output (one device->host word per step)
            if loc["offsets"] is None or loc["offsets"].numel() < total:
                loc["offsets"] = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
            loc["total"] = total
            ix.locate_walk_device(npats, d_res[0].data_ptr(), d_ostarts.data_ptr(), total, loc["offsets"].data_ptr(), stream)
        if world > 1:
            dist.gather(d_res, gather_list, dst=0)   # RCCL over xGMI: the only collective on the path

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ix.kernel_time_reset()
    ix.kernel_time_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ix.kernel_time_enable(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms, kern_n = ix.kernel_time("count")
    loc_ms, loc_n = ix.kernel_time("locate")

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    first = d_res[0].cpu().numpy()
    last = d_res[1].cpu().numpy()
    value = world * npats * args.steps / elapsed

    # ---- CPU baseline + bit-exact check on a bounded sample of the same batch (rank 0 only)
    from oracle import pyoracle as po
    cpu = None
    counters = po.Counters()
    sample = min(args.cpu_sample, npats)
    parity = None
    if sample > 0:
        o = po.Oracle(index_path)
        s_plen, s_starts = plen[:sample], starts[:sample]
        s_flat = flat[:int(starts[sample - 1] + plen[sample - 1])]
        nthr = min(64, os.cpu_count() or 1)
        t0 = time.perf_counter()
        of, ol = o.count_flat(s_plen, s_flat, s_starts, threads=nthr, counters=counters)
        port_mt_s = time.perf_counter() - t0
        parity = bool(np.array_equal(of, first[:sample]) and np.array_equal(ol, last[:sample]))
        assert parity, "GPU results differ from the oracle on the sample"
        loc_counters = po.Counters()
        if locate:
            g_noccs = d_noccs.cpu().numpy()
            g_ost = d_ostarts.cpu().numpy()
            g_offs = loc["offsets"][:loc["total"]].cpu().numpy()
            on, oo = o.locate_flat(s_plen, s_flat, s_starts, args.max_occs, threads=nthr, counters=loc_counters)
            lp = bool(np.array_equal(on, g_noccs[:sample]) and np.array_equal(oo, g_offs[:g_ost[sample]]))
            assert lp, "GPU locate results differ from the oracle on the sample"
        if po.have_ref():
            with tempfile.TemporaryDirectory() as td:
                pf, rf = os.path.join(td, "p.fpat"), os.path.join(td, "r.bin")
                po.write_fpat_flat(pf, s_plen, s_flat)
                out = subprocess.run([po.REF_TOOL, "bench", index_path, pf, "locate" if locate else "count",
                                      str(args.max_occs), "1", "1", rf],
                                     check=True, stdout=subprocess.PIPE).stdout.decode()
                rj = json.loads(out.strip().splitlines()[-1])
                if locate:
                    ref_ok = int(rj["results"]) == int(g_ost[sample])   # same number of located rows; offsets are checked vs the oracle
                else:
                    ref = np.fromfile(rf, dtype=np.int64)
                    ref_ok = bool(np.array_equal(ref[:sample], first[:sample]) and np.array_equal(ref[sample:], last[:sample]))
                assert ref_ok, "GPU results differ from the genuine reference on the sample"
            cpu = {"value": rj["patterns_per_s"], "unit": "patterns/s", "cores": 1, "kind": "reference",
                   "sample": f"first {sample} patterns of the batch, femto {'parallel_locate' if locate else 'parallel_count'} (1 worker thread = the "
                             f"reference's default), index in page cache, 1 warm-up + 1 timed pass",
                   "bit_exact_vs_gpu": ref_ok,
                   "port_all_cores": {"value": sample / port_mt_s, "threads": nthr}}
        else:
            t0 = time.perf_counter()
            o.count_flat(s_plen, s_flat, s_starts, threads=1)
            cpu = {"value": sample / (time.perf_counter() - t0), "unit": "patterns/s", "cores": 1, "kind": "port",
                   "sample": f"first {sample} of the batch's 20-mers, oracle/femto_oracle.c single thread",
                   "bit_exact_vs_gpu": parity,
                   "port_all_cores": {"value": sample / port_mt_s, "threads": nthr}}

    # ---- roofline of the dominant kernel (count_kernel): algorithmic bytes per launch / kernel time
    c = counters.asdict()
    roof = None
    if sample > 0 and kern_n > 0:
        # bytes = N_rank*(12 + 64 + S_rank) + N_occ*20, counters from the CPU
        # restatement on the sample, scaled to the launch's pattern count.
        if locate:   # locate_flat re-runs the count: its counters cover count + walk
            c = loc_counters.asdict()
        alg_sample = c["n_rank"] * (12 + 64) + c["s_bytes"] + c["n_occ"] * 20 + c["n_mark"] * 8
        if locate:
            kern_ms = kern_ms + loc_ms   # both kernels of the step
        alg_launch = alg_sample * (npats / sample)
        achieved = alg_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                if tj.get("npats") == npats and tj.get("text_log2") == args.text_log2 and not locate:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": {1: "count_kernel_lane", 2: "count_kernel_flat", 0: "count_kernel<32>"}[ix.rank_mode] + ("+locate kernel" if locate else ""),
                "kernel_ms": kern_ms, "locate_kernel_ms": loc_ms if locate else None, "launches_timed": kern_n,
                "algorithmic_bytes_per_launch": alg_launch,
                "per_pattern": {"bseq_rank": c["n_rank"] / sample, "occ": c["n_occ"] / sample,
                                "S_bytes_per_rank": c["s_bytes"] / max(1, c["n_rank"]),
                                "bytes": alg_sample / sample},
                "contract_335B_per_occ_GBs": 335.0 * c["n_occ"] / sample * npats / (kern_ms * 1e-3) / 1e9}

    out = {
        "metric": f"patterns/sec ({'count+locate' if locate else 'count'}) on " + ("1 GiB index" if args.text_log2 == 30 else f"2^{args.text_log2} B index"),
        "value": value, "unit": "patterns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": (f"T_eng(2^{args.text_log2}) femto index (default params), {npats} P_hit lengths 8..64 per GPU, count()+locate(max_occs={args.max_occs})" if eng else
                                f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_hit 20-mers per GPU, count()+locate(max_occs={args.max_occs})" if locate else
                                f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_rand 20-mers per GPU, count()"),
                   "located_rows_per_gpu": loc["total"] if locate else None,
                   "text_bytes": n_text, "patterns_per_gpu": npats, "pattern_len": args.plen, "seed": args.seed,
                   "index": {"rows": int(info.total_length), "blocks": int(info.number_of_blocks), "buckets": int(info.total_buckets),
                             "image_bytes": int(info.image_bytes), "table_bytes": int(info.table_bytes)},
                   "parallelism": f"replicated index, query shards x{world}" + (", RCCL gather to rank 0 per step" if world > 1 else ""),
                   "build_s": build_s, "open_s": open_s},
        "roofline": roof, "cpu_baseline": cpu,
        "gpu_vs_cpu": (value / cpu["value"]) if cpu else None,
        "matched_patterns_frac": float(np.mean(last >= first)),
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

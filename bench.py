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
# (rank mode, direct pipeline?) -> how rocprofv3 names the search / walk kernel (a prefix: template arguments follow)
KERNEL_NAMES = {(4, True): "femto_amd::count_direct_kernel<femto_amd::Pack2Policy, true", (3, True): "femto_amd::count_direct_kernel<femto_amd::PackPolicy, true",
                (1, False): "femto_amd::count_kernel_lane", (0, False): "femto_amd::count_kernel<32>"}
LOCATE_NAMES = {(4, True): "femto_amd::locate_walk_kernel<femto_amd::Pack2Policy>", (3, True): "femto_amd::locate_walk_kernel<femto_amd::PackPolicy>",
                (1, False): "femto_amd::locate_kernel_lane", (0, False): "femto_amd::locate_kernel<32>"}


def kernel_names(ix, direct):
    """names (prefixes) of the kernels the count and the locate timers bracket for this handle"""
    pi = ix.pack_info()
    cn, ln = KERNEL_NAMES[(ix.rank_mode, direct)], LOCATE_NAMES[(ix.rank_mode, direct)]
    if direct and ix.rank_mode == 3 and pi.get("rank_units"):
        cn = "femto_amd::count_direct_kernel<femto_amd::RuPolicy, true"
    if direct and ix.rank_mode == 4 and pi.get("char_rank_lines"):
        cn = "femto_amd::count_direct_kernel<femto_amd::IndPolicy, true"
    if direct:      # locate is fused into the row expansion: offsets from the resident suffix array (1) or by a walk per row (2)
        ln = "femto_amd::plan_rows_kernel<1," if pi.get("sa_full") else "femto_amd::plan_rows_kernel<2,"
    return cn, ln


PMC_REGEX = "count_direct_kernel|locate_walk_kernel|count_kernel|locate_kernel|count_tail_kernel|plan_rows_kernel"


def source_hash():
    """hash of the kernel / host sources: ties a committed PMC file to the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "femto_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args, kname, pack_info, open_opts="", child_env=None):
    """HBM-side bytes per launch of kernel `kname`, measured NOW: two separate `rocprofv3 --pmc` passes over a short child
    run of this script (FETCH_SIZE; WRITE_SIZE + request counters -- never combined with any trace domain).  Per the
    guide (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes -> x2; both are KiB."""
    import csv
    import glob
    import shutil
    if not shutil.which("rocprofv3"):
        return None, None
    base = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "2", "--warmup", "1", "--text-log2", str(args.text_log2),
            "--npats", str(args.npats), "--plen", str(args.plen), "--seed", str(args.seed), "--max-occs", str(args.max_occs),
            "--workload", args.workload, "--workdir", args.workdir] + (["--len-range", args.len_range] if args.len_range else [])
    if open_opts or args.open_opts:
        base += ["--open-opts", open_opts or args.open_opts]
    means = {}
    child_info = None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        # The child opens the index while this process still holds its own, so it sees less free HBM: the depths this
        # process chose for the level table / context table are forced, and the child reports what it built.
        env = dict(os.environ, TMPDIR="/tmp", FEMTO_AMD_BENCH_CHILD_INFO=os.path.join(td, "child.json"))
        env.update(child_env or {})
        if pack_info.get("level_table"):
            env["FEMTO_AMD_KTAB_SYMS"] = str(pack_info["ktab_syms"])
        env["FEMTO_AMD_CTX"] = "1" if pack_info.get("context_table") else "0"
        if pack_info.get("context_table"):
            env["FEMTO_AMD_CTX_SYMS"] = str(pack_info["context_syms"])
        env["FEMTO_AMD_CTX2"] = "1" if pack_info.get("context2_syms") else "0"
        if pack_info.get("context2_syms"):
            env["FEMTO_AMD_CTX2_SYMS"] = str(pack_info["context2_syms"])
        for i, ctrs in enumerate((["FETCH_SIZE"], ["WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum"])):
            out = os.path.join(td, f"p{i}")
            cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-include-regex", PMC_REGEX, "-f", "csv", "-d", out, "-o", "pmc", "--"] + base
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, cwd="/tmp", env=env)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if _same_kernel(kname, r.get("Kernel_Name", "")):
                            means.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            try:
                child_info = json.load(open(env["FEMTO_AMD_BENCH_CHILD_INFO"]))
            except Exception:      # noqa: BLE001
                child_info = None
    keys = ("level_table", "ktab_syms", "sa_full", "isa_full", "char_rank_lines", "context_table", "context_syms", "context2_syms", "rank_units")
    if child_info is None or any(child_info.get(k) != pack_info.get(k) for k in keys):
        log("pmc child built different structures, traffic not used:", child_info)
        return None, None
    if "FETCH_SIZE" not in means or "WRITE_SIZE" not in means:
        return None, None
    m = {k: sum(v) / len(v) for k, v in means.items()}
    traffic = 2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    src = {"how": "live: 2 separate rocprofv3 --pmc passes over a 2-step child run of this script in this very run",
           "FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"], "TCC_EA0_RDREQ": m.get("TCC_EA0_RDREQ_sum"),
           "TCC_EA0_RDREQ_128B": m.get("TCC_EA0_RDREQ_128B_sum"), "dispatches": len(means["FETCH_SIZE"]), "source_hash": source_hash(),
           "child_index": {k: child_info.get(k) for k in keys},
           "formula": "2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-B requests as 64 B) + WRITE_SIZE KiB x 1024"}
    return traffic, src


def _same_kernel(kname, full):
    """rocprofv3's kernel name starts (after an optional 'void ') with the wanted prefix"""
    f = full[5:] if full.startswith("void ") else full
    return f.replace(" ", "").startswith(kname.replace(" ", ""))


def committed_traffic(args, kname, npats):
    """fallback: a committed profiles/latest_pmc.json, accepted only when it was measured on these very sources"""
    try:
        tj = json.load(open(args.traffic_json))
        if (tj.get("source_hash") == source_hash() and tj.get("npats") == npats and tj.get("text_log2") == args.text_log2
                and tj.get("workload") == args.workload and tj.get("kernel") == kname):
            return tj.get("hbm_bytes_per_launch"), {"how": "committed " + os.path.relpath(args.traffic_json, ROOT) + " (same source hash)"}
    except Exception:      # noqa: BLE001
        pass
    return None, None


def cpu_quota():
    """CPUs the container may use on average (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU box shows 256 hardware threads
    but runs this process under a quota of 16 -- host-side stages that use 128 threads are bursts, and back-to-back bursts are
    throttled (the 40-75 ms outliers of the host-pointer path)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:      # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:      # noqa: BLE001
        return None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


ENTRY_BYTES = {"level_table": 8, "context_table": 16, "suffix_array": 8, "isa": 8, "char_rank_lines": 32, "rank_units": 16}     # bytes a lookup USES of the 128-byte line it loads


def roofline_block(ix, direct, b, npats, plen, max_occs, cnt_ms, loc_ms, cnt_n):
    """roofline of the dominant kernel of a timed run (without the PMC traffic): see the comment at its call site"""
    cl, ll, trows = ix.trace_lines(npats, b.d_plen.data_ptr(), b.d_flat.data_ptr(), b.d_starts.data_ptr(), max_occs)
    cr, lr = ix.trace_reads()
    n_sym = int(plen.astype(np.int64).sum())
    stream_count = npats * (4 + 8) + 2 * n_sym + npats * (8 + 8 + 4) + 8 * ((npats + 255) // 256)
    stream_locate = trows * (8 + 8)                    # the row in, its text offset out
    comp_count = 128 * sum(cl.values()) + stream_count
    comp_locate = 128 * sum(ll.values()) + stream_locate
    dominant_is_count = cnt_ms >= loc_ms
    k_ms = cnt_ms if dominant_is_count else loc_ms
    comp = comp_count if dominant_is_count else comp_locate
    lines = cl if dominant_is_count else ll
    # what the kernel USES of those lines: a table / array lookup uses one entry of its line, not 128 bytes
    useful = comp - sum((128 - eb) * lines.get(k, 0) for k, eb in ENTRY_BYTES.items())
    kname = kernel_names(ix, direct)[0 if dominant_is_count else 1]
    achieved = comp / (k_ms * 1e-3) / 1e9
    # the same accounting WITHOUT credit for a line that two patterns of the batch both read (SURVEY 8(d) counts per operation
    # too): 128 B for every line READ + the streamed arrays.  Equal to the compulsory bytes when the structures dwarf the batch
    # (the 57 GB level table: 10 M look-ups touch 9.8 M distinct lines); far above them when a small structure is read many
    # times over by an unsorted batch -- there the distinct-line model is bounded by the structure's SIZE, whatever the kernel does.
    reads = cr if dominant_is_count else lr
    read_bytes = 128 * sum(reads.values()) + (stream_count if dominant_is_count else stream_locate)
    line_reads = {"bytes": read_bytes, "GBs": read_bytes / (k_ms * 1e-3) / 1e9, "frac": read_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "lines_read": reads, "lines_read_per_pattern": sum(reads.values()) / npats,
                  "model": "128 B x every line READ (no credit for lines two patterns share) + streamed arrays"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "line_reads": line_reads,
            "traffic": None, "traffic_source": None, "traffic_GBs": None, "traffic_over_compulsory": None,
            "useful": {"bytes": useful, "GBs": useful / (k_ms * 1e-3) / 1e9, "frac": useful / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "model": "as frac, with table / array / rank lines counted as the entry used of them (8-32 B) instead of 128 B"},
            "kernel": kname, "kernel_ms": k_ms, "launches_timed": cnt_n, "count_kernel_ms": cnt_ms, "locate_kernel_ms": loc_ms,
            "compulsory_bytes_per_launch": comp,
            "compulsory": {"count": {"distinct_lines": cl, "streamed_bytes": stream_count, "bytes": comp_count},
                           "locate": {"distinct_lines": ll, "streamed_bytes": stream_locate, "bytes": comp_locate, "rows": trows}},
            "per_pattern_bytes": comp / npats,
            "bytes_model": "128 B x DISTINCT lines loaded (GPU line trace of the same batch) + arrays streamed once; kernel time from HIP events "
                           "on the launch stream; DESIGN.md section 4"}, kname, k_ms, comp, comp_count + comp_locate


def add_traffic(roof, traffic, traffic_src, k_ms, comp):
    roof["traffic"], roof["traffic_source"] = traffic, traffic_src
    roof["traffic_GBs"] = (traffic / (k_ms * 1e-3) / 1e9) if traffic else None
    roof["traffic_over_compulsory"] = (traffic / comp) if traffic else None


class _EventWork:
    """the .wait() of a gather enqueued on a side stream (same shape as torch's async work handle)"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.ev)     # the compute stream waits; the host does not


class Batch:
    """A pattern batch resident in HBM plus its result buffers."""

    def __init__(self, torch, dev, plen, flat):
        from femto_amd import textgen as tg
        self.plen, self.flat, self.starts = plen, flat, tg.starts_of(plen)
        self.n = len(plen)
        self.d_plen = torch.from_numpy(plen).to(dev)
        self.d_flat = torch.from_numpy(flat.view(np.int16)).to(dev)
        self.d_starts = torch.from_numpy(self.starts).to(dev)
        self.d_res2 = [torch.empty((2, self.n), dtype=torch.int64, device=dev) for _ in range(2)]   # [first; last], double buffered
        self.d_res = self.d_res2[0]
        self.d_wire2 = None       # multi-GPU: the ranges as they travel to rank 0 (see wire())
        self.w_res, self.w_cap, self.w_views, self.w_tmp = None, -1, None, None   # ... or match counts + offsets (wire_results())
        self.d_noccs = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.d_ostarts = torch.empty(self.n + 1, dtype=torch.int64, device=dev)
        self.offsets = None
        self.d_total = torch.zeros(2, dtype=torch.int64, device=dev)
        self.total = 0
        self.torch, self.dev = torch, dev

    def step(self, ix, max_occs, stream, buf=0):
        """one enqueue-only call: count, clamp, prefix sum and the locate walk of every matching row -- a single
        stream-ordered chain on the GPU (femto_amd_locate_device); nothing returns to the host inside a step"""
        self.d_res = self.d_res2[buf]
        if self.offsets is None:
            self.offsets = self.torch.empty(max(1 << 20, self.n // 4), dtype=self.torch.int64, device=self.dev)
        ix.locate_device(self.n, self.d_plen.data_ptr(), self.d_flat.data_ptr(), self.d_starts.data_ptr(), max_occs,
                         self.d_res[0].data_ptr(), self.d_res[1].data_ptr(), self.d_noccs.data_ptr(),
                         self.d_ostarts.data_ptr(), self.offsets.data_ptr(), self.offsets.numel(), self.d_total.data_ptr(), stream)

    def settle(self, ix, max_occs, stream):
        """untimed: run one step, read the row count and grow the offsets buffer until everything fits"""
        while True:
            self.step(ix, max_occs, stream)
            tot = self.d_total.cpu().numpy()
            self.total = int(tot[0])
            if not tot[1]:
                return
            self.offsets = self.torch.empty(int(self.total * 1.25) + 1024, dtype=self.torch.int64, device=self.dev)

    def wire(self, rows, buf=0):
        """The (first,last) ranges of this step in the form that is gathered to rank 0: row numbers of an index with
        fewer than 2^31 rows fit int32 (values -1 .. rows), so they travel as 8 instead of 16 bytes per pattern --
        a lossless narrowing; larger indexes send int64."""
        if rows >= (1 << 31) - 1:
            return self.d_res
        if self.d_wire2 is None:
            self.d_wire2 = [self.torch.empty((2, self.n), dtype=self.torch.int32, device=self.dev) for _ in range(2)]
        self.d_wire2[buf].copy_(self.d_res)
        return self.d_wire2[buf]

    def wire_layout(self, rows, cap, bigcap):
        """byte offsets of the gathered buffer: [rows located i64 | n_big i64 | (pattern, count) i64 pairs x bigcap |
        offsets (i32 below 2^31 - 1 rows, else i64) x cap | counts u8 x n | pad to 8]"""
        esz = 4 if rows < (1 << 31) - 1 else 8
        o_big = 16
        o_off = o_big + 16 * bigcap
        o_cnt = o_off + esz * cap
        nbytes = (o_cnt + self.n + 7) & ~7
        return esz, o_big, o_off, o_cnt, nbytes

    def wire_results(self, rows, cap, buf=0, bigcap=1024, ix=None, stream=0):
        """What north_star calls the results -- the match count of every pattern and the located text offsets -- as ONE
        buffer for the gather.  Match counts travel as one byte each (255 = see the list of (pattern, count) pairs for the
        patterns with 255 matches or more: femto_amd_pack_counts_device), offsets as int32 when the index has fewer than
        2^31 - 1 rows; all lossless.  `cap` / `bigcap` are the same on every rank (max over the ranks + slack, agreed in the
        untimed settle phase).  10 MB per 10 M patterns + 4 B per located row, against 160 MB for the (first,last) ranges."""
        t = self.torch
        esz, o_big, o_off, o_cnt, nbytes = self.wire_layout(rows, cap, bigcap)
        dt = t.int32 if esz == 4 else t.int64
        if self.w_res is None or self.w_cap != (cap, bigcap):
            self.w_res = [t.zeros(nbytes, dtype=t.uint8, device=self.dev) for _ in range(2)]
            self.w_cap = (cap, bigcap)
            self.w_views = []
            for w in self.w_res:
                self.w_views.append((w[:8].view(t.int64), w[8:16].view(t.int64), w[o_big:o_off].view(t.int64),
                                     w[o_off:o_cnt].view(dt), w[o_cnt:o_cnt + self.n]))
        tot, nbig, big, off, cnt = self.w_views[buf]
        if ix is not None and cnt.is_cuda:
            ix.pack_counts_device(self.n, self.d_res[0].data_ptr(), self.d_res[1].data_ptr(), cnt.data_ptr(), big.data_ptr(), bigcap,
                                  nbig.data_ptr(), stream)
        else:      # the same format with torch operators (CPU tensors: the gloo tests)
            c = (self.d_res[1] - self.d_res[0] + 1).clamp_(min=0)
            cnt.copy_(c.clamp(max=255))
            idx = t.nonzero(c >= 255).flatten()
            nbig.fill_(int(idx.numel()))
            k = min(int(idx.numel()), bigcap)
            if k:
                big[0:2 * k:2] = idx[:k]
                big[1:2 * k:2] = c[idx[:k]]
        tot.copy_(self.d_total[:1])
        k = min(cap, self.offsets.numel())
        off[:k].copy_(self.offsets[:k])
        return self.w_res[buf]

    def unwire_results(self, raw, rows, cap, bigcap):
        """(rows located, match counts int64[n], offsets int64[min(rows located, cap)]) from one rank's gathered buffer"""
        esz, o_big, o_off, o_cnt, nbytes = self.wire_layout(rows, cap, bigcap)
        raw = np.ascontiguousarray(raw)
        assert raw.dtype == np.uint8 and raw.size == nbytes
        tot = int(raw[:8].view(np.int64)[0])
        nbig = int(raw[8:16].view(np.int64)[0])
        assert nbig <= bigcap, "more patterns with >= 255 matches than the agreed list holds"
        cnt = raw[o_cnt:o_cnt + self.n].astype(np.int64)
        pairs = raw[o_big:o_big + 16 * nbig].view(np.int64).reshape(-1, 2)
        cnt[pairs[:, 0]] = pairs[:, 1]
        off = raw[o_off:o_cnt].view(np.int32 if esz == 4 else np.int64)[:min(tot, cap)].astype(np.int64)
        return tot, cnt, off


def multi_gpu_extras(args, torch, dist, femto_amd, ix, batch, rank, world, local_rank, dev, backend, native, payload_of, index_path,
                     elapsed, npats, per_rank):
    """N > 1 only, after the timed run: the step again (a) without any gather and (b) with the OTHER gather implementation, so
    one driver run tells search time from gather time and the two gathers apart.  Nothing here has ever run on more than one
    physical GPU before the driver's scaling run, so every part is guarded: a watchdog prints a minimal result line (the
    headline value and what is known so far) and ends the process if a part does not finish -- the headline never depends
    on an extra."""
    import threading
    out = {}
    state = {"phase": "start"}

    def bail():
        if rank == 0:
            line = {"metric": "patterns/sec (count+locate) on " + ("1 GiB index" if args.text_log2 == 30 else f"2^{args.text_log2} B index"),
                    "value": world * npats * args.steps / elapsed, "unit": "patterns/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                    "config": {"workload": f"{npats} patterns per GPU, count()+locate(max_occs={args.max_occs}), layout {args.layout}",
                               "per_rank": per_rank, "note": f"multi-GPU extra '{state['phase']}' did not finish within its time limit: "
                                                             "minimal line, no roofline / cpu_baseline"},
                    "roofline": None, "cpu_baseline": None, "extra": dict(out, timeout=state["phase"])}
            print(json.dumps(line), flush=True)
        os._exit(0)

    def timed(name, fn, steps=5, limit_s=90):
        state["phase"] = name
        t = threading.Timer(limit_s, bail)
        t.daemon = True
        t.start()
        try:
            fn()                                   # warm-up
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
            tm = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            out[name] = {"value": world * npats * steps / float(tm.item()), "unit": "patterns/s", "ms_per_step": 1e3 * float(tm.item()) / steps,
                         "steps": steps}
        except Exception as ex:      # noqa: BLE001
            out[name] = {"error": repr(ex)}
        finally:
            t.cancel()

    stream = torch.cuda.current_stream().cuda_stream
    timed("search_only_no_gather", lambda: batch.step(ix, args.max_occs, stream, 0))
    if backend == "nccl":
        payload = payload_of(0)
        nbytes = payload.numel() * payload.element_size()
        if native:      # the timed run used femto_amd_comm_gather: now torch.distributed.gather (RCCL)
            lists = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None

            def other():
                batch.step(ix, args.max_occs, stream, 0)
                dist.gather(payload_of(0), lists, dst=0)
            timed("gather_torch_distributed", other)
        else:           # the timed run used torch.distributed.gather: now the C ABI's grouped ncclSend / ncclRecv
            try:
                state["phase"] = "native_comm_init"
                t = threading.Timer(90, bail)
                t.daemon = True
                t.start()
                ids = [femto_amd.Index.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ix.comm_init(ids[0], world, rank)
                t.cancel()
                recv = torch.empty((world,) + tuple(payload.shape), dtype=payload.dtype, device=dev) if rank == 0 else None

                def other():
                    batch.step(ix, args.max_occs, stream, 0)
                    p = payload_of(0)
                    ix.comm_gather(p.data_ptr(), recv.data_ptr() if rank == 0 else 0, nbytes, 0, stream)
                timed("gather_native_ncclSendRecv", other)
                out["native_comm"] = ix.comm_info()
            except Exception as ex:      # noqa: BLE001
                out["gather_native_ncclSendRecv"] = {"error": repr(ex)}
        out["gather_payload_bytes_per_rank"] = int(nbytes)
    state["phase"] = "done"
    return out if rank == 0 else None


def timed_steps(torch, ix, b, max_occs, stream, steps, warm=2):
    """`steps` timed passes of batch `b` on handle `ix` (inputs and outputs resident): wall seconds, count / locate kernel ms"""
    b.settle(ix, max_occs, stream)
    for _ in range(warm):
        b.step(ix, max_occs, stream)
    torch.cuda.synchronize()
    ix.kernel_time_reset()
    ix.kernel_time_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step(ix, max_occs, stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ix.kernel_time_enable(False)
    return el, ix.kernel_time("count"), ix.kernel_time("locate")


def reference_format_block(cd, csub, npats, k_ms):
    """SURVEY 8(d)'s byte formula on the REFERENCE's own operation counts for this batch (oracle counters on `csub` patterns):
    N_rank x (12 + 64) + S bytes consumed + N_occ x 20 + N_mark x 8 -- what femto's algorithm would move on femto's format."""
    b = (cd["n_rank"] * (12 + 64) + cd["s_bytes"] + cd["n_occ"] * 20 + cd["n_mark"] * 8) / csub
    gbs = b * npats / (k_ms * 1e-3) / 1e9 if k_ms else None
    return {"bytes_per_pattern": b, "GBs": gbs, "x_peak": (gbs / HBM_PEAK_GBS) if gbs else None,
            "formula": f"SURVEY 8(d): N_rank*(12+64) + S_bytes + N_occ*20 + N_mark*8, oracle counters on {csub} patterns"}


def budget_extra(args, torch, femto_amd, tg, dev, local_rank, index_path, text_path, batch, plen, ref_results, stream, n_text):
    """The footprint-bounded open (round-3 verdict, task 1): the SAME index with hbm_budget_bytes = 4 x text bytes -- packed lines,
    rank units, sampled marks and the level table the rest pays for; no dense suffix arrays, no text -- on the headline batch
    (random 20-mers) and on 20-mers sampled from the text, with its own roofline block and live PMC traffic."""
    npats = args.npats
    budget = 4 * n_text
    opts = {"hbm_budget_bytes": budget}
    bix = femto_amd.Index(index_path, device=local_rank, options=opts)
    out = {"what": f"same index opened with femto_amd_open_opts(hbm_budget_bytes = 4 x text = {budget}): search steps on the rank units / packed lines, "
                   "no dense suffix arrays, no text tail", "structures": bix.structures(), "index": bix.pack_info()}
    try:
        steps = max(5, args.steps)
        el, (c_ms, c_n), (l_ms, _) = timed_steps(torch, bix, batch, args.max_occs, stream, steps)
        first, last, noccs, ost, offs = ref_results
        same = bool(np.array_equal(batch.d_res[0].cpu().numpy(), first) and np.array_equal(batch.d_res[1].cpu().numpy(), last)
                    and np.array_equal(batch.d_noccs.cpu().numpy(), noccs) and np.array_equal(batch.offsets[:batch.total].cpu().numpy(), offs))
        assert same, "budget-bounded handle: results differ from the headline handle's"
        out.update({"workload": f"{npats} P_rand 20-mers, count()+locate(max_occs={args.max_occs})", "value": npats * steps / el, "unit": "patterns/s",
                    "ms_per_step": 1e3 * el / steps, "steps": steps, "count_kernel_ms": c_ms, "locate_kernel_ms": l_ms,
                    "equal_to_headline_results": same})
        roof, kname, k_ms, comp, _ = roofline_block(bix, True, batch, npats, plen, args.max_occs, c_ms, l_ms, c_n)
        info = bix.pack_info()
        out["roofline"] = roof
        # 20-mers sampled from the text: every pattern runs all its steps and is located by a walk to the next derived mark
        text = np.load(text_path, mmap_mode="r")
        hp, hf = tg.p_hit(args.plen, args.plen, npats, args.seed + 2000, np.asarray(text))
        del text
        hb = Batch(torch, dev, hp, hf)
        hel, (hc_ms, hc_n), (hl_ms, _) = timed_steps(torch, bix, hb, args.max_occs, stream, 3)
        hroof, _, _, _, _ = roofline_block(bix, True, hb, npats, hp, args.max_occs, hc_ms, hl_ms, hc_n)
        out["p_hit"] = {"workload": f"{npats} 20-mers sampled from the text, count()+locate(max_occs={args.max_occs})", "value": npats * 3 / hel,
                        "unit": "patterns/s", "ms_per_step": 1e3 * hel / 3, "located_rows": hb.total, "count_kernel_ms": hc_ms,
                        "locate_kernel_ms": hl_ms, "roofline": {k: hroof[k] for k in ("achieved", "frac", "kernel", "kernel_ms", "compulsory_bytes_per_launch", "line_reads")}}
        del hb
        bix.close()
        bix = None
        if args.pmc != "off":
            try:
                tr, trs = pmc_traffic(args, kname, info, open_opts=f"hbm_budget_bytes={budget}")
                add_traffic(roof, tr, trs, k_ms, comp)
            except Exception as ex:      # noqa: BLE001
                log("budget pmc pass failed:", repr(ex))
    except Exception as ex:      # noqa: BLE001
        out["error"] = repr(ex)
    finally:
        if bix is not None:
            bix.close()
    return out


def mode1_extra(args, torch, ix, batch, ref_results, stream, cd_count, csub):
    """SURVEY 8(d)'s own kernel family: femto's wavelet tree itself (mode 1: one lane per pattern on the derived segment lines
    of femto's RLE / literal sequences, batch ordered by suffix).  The only family 8(d)'s byte formula describes."""
    out = {"what": "the headline batch through rank mode 1 (count_kernel_lane / locate_kernel_lane on femto's own wavelet tree, suffix-ordered batch)"}
    old = ix.rank_mode
    try:
        ix.set_rank_mode(1)
        el, (c_ms, c_n), (l_ms, _) = timed_steps(torch, ix, batch, args.max_occs, stream, 3, warm=1)
        first, last, noccs, ost, offs = ref_results
        same = bool(np.array_equal(batch.d_res[0].cpu().numpy(), first) and np.array_equal(batch.d_res[1].cpu().numpy(), last)
                    and np.array_equal(batch.offsets[:batch.total].cpu().numpy(), offs))
        assert same, "mode 1: results differ from the packed path's"
        out.update({"value": args.npats * 3 / el, "unit": "patterns/s", "ms_per_step": 1e3 * el / 3, "count_kernel_ms": c_ms,
                    "locate_kernel_ms": l_ms, "equal_to_headline_results": same, "kernel": "femto_amd::count_kernel_lane"})
        if cd_count:
            rf = reference_format_block(cd_count, csub, args.npats, c_ms)
            out["roofline"] = {"bound": "hbm", "achieved": rf["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rf["x_peak"], "kernel_ms": c_ms,
                               "bytes_per_pattern": rf["bytes_per_pattern"], "bytes_model": rf["formula"] + " (count only: the kernel timed is the search)"}
    except Exception as ex:      # noqa: BLE001
        out["error"] = repr(ex)
    finally:
        ix.set_rank_mode(old)
    if "roofline" in out and args.pmc != "off":
        try:
            tr, trs = pmc_traffic(args, "femto_amd::count_kernel_lane", ix.pack_info(), child_env={"FEMTO_AMD_RANK_MODE": "lane"})
            if tr:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_GBs"] = tr / (out["count_kernel_ms"] * 1e-3) / 1e9
                out["roofline"]["traffic_source"] = trs
        except Exception as ex:      # noqa: BLE001
            log("mode-1 pmc pass failed:", repr(ex))
    return out


def shim_extras(args, index_path, plen, flat, first, last, located_rows):
    """The drop-in as a femto caller sees it: oracle/_ref/ref_tool_amd -- our driver making query_tool.c's calls, linked with
    integration/femto_amd_shim.c -- runs parallel_count / parallel_locate (alpha_t** pointer-per-pattern arrays in pageable
    memory, femto.c:275-386) on the headline batch; the process opens the index on the GPU itself (untimed warm-up pass)."""
    from oracle import pyoracle as po
    res = {}
    if not os.path.exists(po.REF_TOOL_AMD):
        return {"shim_parallel_count": {"error": "oracle/_ref/ref_tool_amd not built"}}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        pf, rf = os.path.join(td, "p.fpat"), os.path.join(td, "r.bin")
        po.write_fpat_flat(pf, plen, flat)
        for name, mode in (("shim_parallel_count", "count"), ("shim_parallel_locate", "locate")):
            try:
                reps = 8
                o_ = subprocess.run([po.REF_TOOL_AMD, "bench", index_path, pf, mode, str(args.max_occs), "1", str(reps)] + ([rf] if mode == "count" else []),
                                    check=True, stdout=subprocess.PIPE, timeout=300).stdout.decode()
                tj = json.loads(o_.strip().splitlines()[-1])
                e = {"what": f"parallel_{mode} of femto_internal.h through integration/femto_amd_shim.c (ref_tool_amd bench: alpha_t** patterns, pageable memory, "
                             "results in the caller's arrays" + ("; offsets[i] malloc()ed per matching pattern" if mode == "locate" else "") +
                             f"), 1 warm-up + {reps} timed calls in a fresh process: value = mean, best = fastest call (the first calls after the index opens run 2-4x slower: "
                             "the staging threads have gone to sleep while the caller freed the previous results)",
                     "value": len(plen) / tj["mean_s"], "best": len(plen) / tj["best_s"], "unit": "patterns/s", "ms": 1e3 * tj["mean_s"], "best_ms": 1e3 * tj["best_s"]}
                if mode == "count":
                    r = np.fromfile(rf, dtype=np.int64)
                    e["equal_to_device_path"] = bool(np.array_equal(r[:len(plen)], first) and np.array_equal(r[len(plen):], last))
                else:
                    e["results"] = int(tj["results"])
                    e["equal_to_device_path"] = bool(int(tj["results"]) == int(located_rows))
                res[name] = e
            except Exception as ex:      # noqa: BLE001
                res[name] = {"error": repr(ex)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
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
    ap.add_argument("--gather", default="torch", choices=["torch", "native"],
                    help="N > 1: torch = torch.distributed.gather (RCCL); native = the library's own grouped ncclSend/ncclRecv "
                         "(femto_amd_comm_gather), its id broadcast through torch.distributed")
    ap.add_argument("--results", default="counts", choices=["counts", "ranges"],
                    help="N > 1: what is gathered to rank 0 every step: counts = match count of every pattern + the located offsets "
                         "(north_star's 'results'), ranges = the (first,last) row ranges of parallel_count")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary P_hit line (N=1 default workload only)")
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "latest_pmc.json"))
    ap.add_argument("--open-opts", default="", help="experiments: femto_amd_options_t fields of the headline handle, 'name=value,...' "
                                                    "(e.g. hbm_budget_bytes=4294967296,rank_units=0)")
    args = ap.parse_args()
    open_opts = {k: int(v) for k, v in (kv.split("=") for kv in args.open_opts.split(",") if kv)} or None

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
    direct = ix.rank_mode in (3, 4)      # the caller-order pipeline (direct_kernels.hip.hpp)
    if args.pmc_child:      # the short run the PMC passes of pmc_traffic() profile: same index, same batch, a few steps
        if os.environ.get("FEMTO_AMD_BENCH_CHILD_INFO"):
            with open(os.environ["FEMTO_AMD_BENCH_CHILD_INFO"], "w") as fh:
                json.dump(ix.pack_info(), fh)
        st = torch.cuda.current_stream().cuda_stream
        batch.settle(ix, args.max_occs, st)
        for _ in range(args.warmup + args.steps):
            batch.step(ix, args.max_occs, st)
        torch.cuda.synchronize()
        ix.close()
        return
    cap, bigcap = 0, 1024
    if world > 1 and args.results == "counts":     # untimed: every rank's row total, the common capacity of the gathered offsets
        batch.settle(ix, args.max_occs, torch.cuda.current_stream().cuda_stream)
        tcap = torch.tensor([batch.total], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
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
                "located_rows": batch.total, "own_wall_s": own_elapsed, "world_size_seen": dist.get_world_size(),
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

    first = batch.d_res[0].cpu().numpy()
    last = batch.d_res[1].cpu().numpy()
    g_noccs = batch.d_noccs.cpu().numpy()
    g_ost = batch.d_ostarts.cpu().numpy()
    g_offs = batch.offsets[:batch.total].cpu().numpy()
    located_rows = batch.total
    value = world * npats * args.steps / elapsed
    gathered_ok = None
    if world > 1 and args.results == "counts" and counter["k"]:
        # what arrived on rank 0 in the last step: every rank's buffer decodes, and rank 0's own slot equals its local results
        b_last = (counter["k"] - 1) & 1
        slots = recv_native[b_last] if native else gather_lists[b_last]
        gathered_ok = True
        for r in range(world):
            tot_r, cnt_r, off_r = batch.unwire_results(slots[r].cpu().numpy(), info.total_length, cap, bigcap)
            gathered_ok = gathered_ok and tot_r >= 0 and int(cnt_r.min()) >= 0
            if r == 0:
                gathered_ok = (gathered_ok and tot_r == batch.total and np.array_equal(cnt_r, np.maximum(last - first + 1, 0))
                               and np.array_equal(off_r, g_offs[:len(off_r)]))
        assert gathered_ok, "the gathered results do not decode to rank 0's own results"

    # ---- secondary line: every pattern occurs and is located (same index, P_hit 20-mers)
    extra = None
    if want_extra:
        text = np.load(text_path, mmap_mode="r")
        hp, hf = tg.p_hit(args.plen, args.plen, npats, args.seed + 2000, np.asarray(text))
        del text
        hb = Batch(torch, dev, hp, hf)
        hb.settle(ix, args.max_occs, stream)
        for _ in range(2):
            hb.step(ix, args.max_occs, stream)
        torch.cuda.synchronize()
        ix.kernel_time_reset()
        ix.kernel_time_enable(True)
        t0 = time.perf_counter()
        for _ in range(3):
            hb.step(ix, args.max_occs, stream)
        torch.cuda.synchronize()
        he = time.perf_counter() - t0
        ix.kernel_time_enable(False)
        extra = {"p_hit_count_locate": {"workload": f"{npats} 20-mers sampled from the text, count()+locate(max_occs={args.max_occs})",
                                        "value": npats * 3 / he, "unit": "patterns/s", "ms_per_step": 1e3 * he / 3,
                                        "located_rows": hb.total, "count_kernel_ms": ix.kernel_time("count")[0],
                                        "locate_kernel_ms": ix.kernel_time("locate")[0]}}
        del hb
        # The headline batch once more as 64-bit KEYS (femto_amd_pack_keys_device: 3 bits per DNA symbol, packed once, untimed --
        # the form a caller that keeps its patterns in HBM would hold them in) with the ranges returned as int32 pairs:
        # 28 instead of 80 bytes streamed per pattern around the same search.  Same results, checked below; an `extra`
        # line, never the headline (whose input is the reference's own alpha_t symbols).
        try:
            d_keys = torch.empty(npats, dtype=torch.int64, device=dev)
            d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
            ix.pack_keys_device(npats, batch.d_plen.data_ptr(), batch.d_flat.data_ptr(), batch.d_starts.data_ptr(), d_keys.data_ptr(),
                                d_bad.data_ptr(), stream)
            torch.cuda.synchronize()
            assert int(d_bad.item()) == 0, "a pattern of the batch does not fit a key"
            k_r32 = torch.empty(2 * npats, dtype=torch.int32, device=dev)
            k_noccs = torch.empty(npats, dtype=torch.int32, device=dev)
            k_ost = torch.empty(npats + 1, dtype=torch.int64, device=dev)
            k_offs = torch.empty(batch.offsets.numel(), dtype=torch.int64, device=dev)
            k_total = torch.zeros(2, dtype=torch.int64, device=dev)

            def kstep():
                ix.locate_keys_device(npats, d_keys.data_ptr(), args.max_occs, k_r32.data_ptr(), 0, 0, k_noccs.data_ptr(), k_ost.data_ptr(),
                                      k_offs.data_ptr(), k_offs.numel(), k_total.data_ptr(), stream)
            for _ in range(3):
                kstep()
            torch.cuda.synchronize()
            ix.kernel_time_reset()
            ix.kernel_time_enable(True)
            ksteps = max(3, args.steps)
            t0 = time.perf_counter()
            for _ in range(ksteps):
                kstep()
            torch.cuda.synchronize()
            ke = time.perf_counter() - t0
            ix.kernel_time_enable(False)
            pairs = k_r32.cpu().numpy().reshape(npats, 2)
            same = bool(np.array_equal(pairs[:, 0], first) and np.array_equal(pairs[:, 1], last) and np.array_equal(k_noccs.cpu().numpy(), g_noccs)
                        and np.array_equal(k_ost.cpu().numpy(), g_ost) and np.array_equal(k_offs[:batch.total].cpu().numpy(), g_offs))
            assert same, "the key path's results differ from the symbol path's"
            extra["compact_keys_count_locate"] = {
                "what": "the headline batch as 64-bit keys in, int32 (first,last) pairs + row counts + located offsets out (femto_amd_locate_keys_device)",
                "value": npats * ksteps / ke, "unit": "patterns/s", "ms_per_step": 1e3 * ke / ksteps, "steps": ksteps,
                "count_kernel_ms": ix.kernel_time("count")[0], "streamed_bytes_per_pattern": 28, "equal_to_symbol_path": same}
            del d_keys, k_r32, k_noccs, k_ost, k_offs
        except Exception as ex:      # noqa: BLE001
            extra["compact_keys_count_locate"] = {"error": repr(ex)}
        # Automaton search (do_regexp_query for a BATCH of automata, one workgroup each): random DNA motifs with
        # classes / alternations / optional symbols on the headline index, the genuine reference (one thread, its only mode)
        # timed on a sample of the same automata and its result lists compared with the GPU's.
        try:
            from oracle import pyoracle as po_rx
            rrng = np.random.Generator(np.random.PCG64(args.seed + 77))

            def motif(k):
                out = b""
                for _ in range(k):
                    r = rrng.random()
                    if r < 0.70:
                        out += bytes([b"ACGT"[rrng.integers(0, 4)]])
                    elif r < 0.85:
                        out += b"[" + bytes(sorted(set(b"ACGT"[i] for i in rrng.integers(0, 4, 2)))) + b"]"
                    elif r < 0.93:
                        out += (b"(" + bytes(b"ACGT"[i] for i in rrng.integers(0, 4, 2)) + b"|" + bytes(b"ACGT"[i] for i in rrng.integers(0, 4, 2)) + b")")
                    else:
                        out += bytes([b"ACGT"[rrng.integers(0, 4)]]) + b"?"
                return out
            n_rx = 5000
            nfas = [femto_amd.Nfa.from_regex(motif(int(rrng.integers(14, 19)))) for _ in range(n_rx)]
            ix.nfa_search_batch(nfas[:128], max_results=1 << 22)
            t0 = time.perf_counter()
            r_start, r_first, r_last, r_len, r_cost, r_status = ix.nfa_search_batch(nfas, max_results=1 << 24)
            rx_s = time.perf_counter() - t0
            rx = {"what": f"{n_rx} random DNA motifs of 14-18 terms (classes, alternations, optional symbols) as ONE femto_amd_nfa_search_batch call "
                          "(automata compiled beforehand; upload, search and result sort inside the timed call)",
                  "value": n_rx / rx_s, "unit": "automata/s", "ms": 1e3 * rx_s, "result_ranges": int(len(r_first)),
                  "not_ok": int((r_status != 0).sum())}
            if po_rx.have_ref():
                m = 100
                with tempfile.TemporaryDirectory(dir="/tmp") as td:
                    t0 = time.perf_counter()
                    ref_rx = po_rx.ref_regexp_nfa(index_path, nfas[:m], td)
                    ref_s = time.perf_counter() - t0
                same = all(rr[0] == int(r_status[i]) and np.array_equal(rr[1], r_first[r_start[i]:r_start[i + 1]])
                           and np.array_equal(rr[2], r_last[r_start[i]:r_start[i + 1]]) and np.array_equal(rr[3], r_len[r_start[i]:r_start[i + 1]])
                           and np.array_equal(rr[4], r_cost[r_start[i]:r_start[i + 1]]) for i, rr in enumerate(ref_rx))
                assert same, "automaton search: GPU result lists differ from the genuine do_regexp_query"
                rx["cpu_baseline"] = {"value": m / ref_s, "unit": "automata/s", "kind": "reference", "cores": 1,
                                      "sample": f"the first {m} automata through setup_regexp_query_take_nfa + do_regexp_query (ref_tool regexp_nfa, "
                                                "process start and index open included)", "identical_results": True}
            extra["regexp_batch"] = rx
        except Exception as ex:      # noqa: BLE001
            extra["regexp_batch"] = {"error": repr(ex)}
        # PCIe-inclusive rate of the host-pointer entry point (patterns and results in pageable host memory):
        # never the headline value, reported for the drop-in caller's benefit
        hf_ = np.zeros(npats, dtype=np.int64) + 1      # touched: the call is timed, not the first-touch page faults
        hl_ = np.zeros(npats, dtype=np.int64) + 1
        hstarts = np.ascontiguousarray(batch.starts, dtype=np.int64)
        L = femto_amd.lib()
        hts = []
        for _ in range(6):                             # first call allocates the pinned staging buffers
            t0 = time.perf_counter()
            rc = L.femto_amd_count_flat(ix.handle, npats, plen.ctypes.data, flat.ctypes.data, hstarts.ctypes.data,
                                        hf_.ctypes.data, hl_.ctypes.data)
            hts.append(time.perf_counter() - t0)
            assert rc == 0
        hs = min(hts[1:])
        extra["host_pointer_count"] = {"what": "femto_amd_count_flat on the same 10M-pattern batch, pageable host arrays in and out (staging threads + PCIe + kernels, pipelined in "
                                               "1M-pattern stages, three in flight); value = fastest of 5 calls after a warm-up (single calls run longer once the spinning "
                                               "staging threads have spent the container's CPU quota -- cgroup_cpu_quota CPUs on average: mean_ms)",
                                       "value": npats / hs, "unit": "patterns/s", "ms": 1e3 * hs, "mean_ms": 1e3 * sum(hts[1:]) / len(hts[1:]),
                                       "host_hardware_threads": os.cpu_count(), "cgroup_cpu_quota": cpu_quota(),
                                       "equal_to_device_path": bool(np.array_equal(hf_, first) and np.array_equal(hl_, last))}
        del hf_, hl_
        extra.update(shim_extras(args, index_path, plen, flat, first, last, located_rows))
    # ---- CPU baseline + bit-exact check on a bounded sample of the same batch (rank 0 only)
    from oracle import pyoracle as po
    cpu = None
    ref_work = None
    cd_count, csub = None, 0
    host_cores = os.cpu_count() or 1
    sample = min(args.cpu_sample, npats)
    if sample > 0:
        o = po.Oracle(index_path)
        s_plen, s_starts = plen[:sample], batch.starts[:sample]
        s_flat = flat[:int(s_starts[-1] + s_plen[-1])]
        nthr = min(64, host_cores)
        t0 = time.perf_counter()
        of, ol = o.count_flat(s_plen, s_flat, s_starts, threads=nthr)
        on, oo = o.locate_flat(s_plen, s_flat, s_starts, args.max_occs, threads=nthr)
        port_mt_s = time.perf_counter() - t0
        assert np.array_equal(of, first[:sample]) and np.array_equal(ol, last[:sample]), "GPU count differs from the oracle"
        assert np.array_equal(on, g_noccs[:sample]) and np.array_equal(oo, g_offs[:g_ost[sample]]), "GPU locate differs from the oracle"
        # "always report Occ/s alongside patterns/s" -- what the REFERENCE's algorithm does for these patterns
        # (the restatement's deterministic counters on a small single-thread sample), scaled to the measured rate
        csub = min(sample, 20_000)
        ctr = po.Counters()
        o.locate_flat(s_plen[:csub], s_flat, s_starts[:csub], args.max_occs, threads=1, counters=ctr)
        cd = ctr.asdict()
        ctr_c = po.Counters()
        o.count_flat(s_plen[:csub], s_flat, s_starts[:csub], threads=1, counters=ctr_c)
        cd_count = ctr_c.asdict()
        ref_work = {"sample": csub, "occ_per_pattern": cd["n_occ"] / csub, "bseq_rank_per_pattern": cd["n_rank"] / csub,
                    "lf_steps_per_pattern": cd["n_lf"] / csub, "mark_reads_per_pattern": cd["n_mark"] / csub,
                    "occ_per_s": value * cd["n_occ"] / csub,
                    "what": "operation counts of femto's own algorithm for this batch (oracle counters); occ_per_s = value x occ_per_pattern"}
        if po.have_ref():
            rsample = min(sample, args.ref_sample)       # ~5 s per pass at the reference's ~19 k patterns/s
            with tempfile.TemporaryDirectory() as td:
                pf, rf = os.path.join(td, "p.fpat"), os.path.join(td, "r.bin")
                po.write_fpat_flat(pf, s_plen[:rsample], s_flat[:int(s_starts[rsample - 1] + s_plen[rsample - 1])])
                out = subprocess.run([po.REF_TOOL, "bench", index_path, pf, "locate", str(args.max_occs), "1", "3"],
                                     check=True, stdout=subprocess.PIPE).stdout.decode()
                rj = json.loads(out.strip().splitlines()[-1])
                assert int(rj["results"]) == int(g_ost[rsample]), "located-row count differs from the genuine reference"
                sub = min(rsample, 50_000)   # direct range check against the reference's parallel_count
                po.write_fpat_flat(pf, s_plen[:sub], s_flat[:int(s_starts[sub - 1] + s_plen[sub - 1])])
                subprocess.run([po.REF_TOOL, "count", index_path, pf, rf], check=True, stdout=subprocess.PIPE)
                ref = np.fromfile(rf, dtype=np.int64)
                assert np.array_equal(ref[:sub], first[:sub]) and np.array_equal(ref[sub:], last[:sub]), \
                    "GPU ranges differ from the genuine reference"
                # the reference with num_threads = 2 / 4 / 8 (server_settings_t, src/main/server.c:3484-3602; its
                # default is forced to 1 at :3597) next to the 1-thread figure -- a third of the sample, 2 timed passes each
                ref_threads = {}
                tsample = max(1000, rsample // 3)
                po.write_fpat_flat(pf, s_plen[:tsample], s_flat[:int(s_starts[tsample - 1] + s_plen[tsample - 1])])
                for nthr_ref in (2, 4, 8):
                    try:
                        o_ = subprocess.run([po.REF_TOOL, "bench", index_path, pf, "locate", str(args.max_occs), str(nthr_ref), "2"],
                                            check=True, stdout=subprocess.PIPE, timeout=120).stdout.decode()
                        tj = json.loads(o_.strip().splitlines()[-1])
                        ref_threads[str(nthr_ref)] = {"value": tsample / tj["mean_s"], "best": tsample / tj["best_s"], "sample": tsample}
                    except Exception as ex:      # noqa: BLE001
                        ref_threads[str(nthr_ref)] = {"error": repr(ex)}
            cpu = {"value": rsample / rj["mean_s"], "unit": "patterns/s", "cores": 1, "host_cores": host_cores, "kind": "reference",
                   "sample": f"first {rsample} patterns of the batch through femto's parallel_locate (count + locate, max_occs "
                             f"{args.max_occs}; 1 worker thread = the reference's hard-wired default, src/main/server.c:3597), index in "
                             f"page cache, 1 warm-up + 3 timed passes (mean; best {rsample / rj['best_s']:.0f} patterns/s)",
                   "bit_exact_vs_gpu": True,
                   "reference_num_threads": ref_threads,
                   "port_all_cores": {"value": sample / port_mt_s, "threads": nthr, "host_cores": host_cores,
                                      "what": f"oracle/femto_oracle.c count+locate on the first {sample} patterns"}}
        else:
            t0 = time.perf_counter()
            o.locate_flat(s_plen, s_flat, s_starts, args.max_occs, threads=1)
            cpu = {"value": sample / (time.perf_counter() - t0), "unit": "patterns/s", "cores": 1, "host_cores": host_cores, "kind": "port",
                   "sample": f"first {sample} patterns of the batch, oracle/femto_oracle.c count+locate, single thread",
                   "bit_exact_vs_gpu": True,
                   "port_all_cores": {"value": sample / port_mt_s, "threads": nthr, "host_cores": host_cores}}

    # ---- roofline of the dominant kernel.  achieved = COMPULSORY bytes per launch / average kernel duration, where the
    # compulsory bytes are what the launch must move at least once: 128 B for every DISTINCT line of a derived array it
    # loads (counted on the GPU by a traced run of the same batch, femto_amd_trace_lines) + the arrays it streams
    # exactly once (pattern lengths / starts / symbols in, ranges and row counts out).  frac <= 1 by construction.
    roof = None
    if cnt_n > 0:
        try:
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
        res_ref = (first, last, g_noccs, g_ost, g_offs)
        extra["budget4x"] = budget_extra(args, torch, femto_amd, tg, dev, local_rank, index_path, text_path, batch, plen, res_ref, stream, n_text)
        extra["mode1_wavelet_tree"] = mode1_extra(args, torch, ix, batch, res_ref, stream, cd_count, csub)
        if roof is not None:      # compact copies inside the block the driver keeps
            b4 = extra["budget4x"]
            br = b4.get("roofline") or {}
            roof["budget4x"] = {"value": b4.get("value"), "ms_per_step": b4.get("ms_per_step"), "kernel_ms": br.get("kernel_ms"),
                                "frac_distinct_lines": br.get("frac"), "frac_line_reads": (br.get("line_reads") or {}).get("frac"),
                                "traffic_GBs": br.get("traffic_GBs"), "hbm_held": (b4.get("structures") or {}).get("hbm_allocated"),
                                "level_table_syms": (b4.get("structures") or {}).get("level_table_syms"),
                                "p_hit_value": (b4.get("p_hit") or {}).get("value"), "error": b4.get("error")}
            m1 = extra["mode1_wavelet_tree"]
            roof["mode1"] = {"value": m1.get("value"), "count_kernel_ms": m1.get("count_kernel_ms"), "frac_reference_format": (m1.get("roofline") or {}).get("frac"),
                             "traffic_GBs": (m1.get("roofline") or {}).get("traffic_GBs"), "error": m1.get("error")}

    # BASELINE configs[2] as an extra line, LAST: the headline index is closed first, so that the sigma~96 index is opened with
    # the whole HBM to budget against (opened next to the 79 GB DNA index its wide context table got the denser, slower layout)
    main_rank_mode, main_pack_info, main_structs = ix.rank_mode, ix.pack_info(), ix.structures()
    if want_extra:
        del batch
        ix.close()
        ix = None
        torch.cuda.empty_cache()
        # BASELINE configs[2] shape: a sigma~96 text of the same size, sampled patterns of length 8..64 (the two-level
        # 16-ary lines, mode 4).  A failure here must not cost the headline line.
        try:
            e_path = os.path.join(args.workdir, f"eng_2p{args.text_log2}_s{args.seed}")
            e_text = tg.t_eng_torch(n_text, args.seed, f"cuda:{local_rank}")
            if not os.path.exists(os.path.join(e_path, "_femto_index")):
                femto_amd.build_index(e_path, [e_text], params=None, infos=["bench"], device=local_rank)
            eix = femto_amd.Index(e_path, device=local_rank)
            ep, ef = tg.p_hit(8, 64, npats, args.seed + 3000, e_text)
            del e_text
            eb = Batch(torch, dev, ep, ef)
            eb.settle(eix, args.max_occs, stream)
            for _ in range(2):
                eb.step(eix, args.max_occs, stream)
            torch.cuda.synchronize()
            eix.kernel_time_reset()
            eix.kernel_time_enable(True)
            t0 = time.perf_counter()
            for _ in range(3):
                eb.step(eix, args.max_occs, stream)
            torch.cuda.synchronize()
            ee = time.perf_counter() - t0
            eix.kernel_time_enable(False)
            extra["cfg3_text96_count_locate"] = {
                "workload": f"T_eng(2^{args.text_log2}) sigma~96 index, {npats} sampled patterns of length 8..64, count()+locate(max_occs={args.max_occs})",
                "rank_mode": {4: "pack2", 3: "pack", 1: "lane", 0: "raw"}[eix.rank_mode],
                "value": npats * 3 / ee, "unit": "patterns/s", "ms_per_step": 1e3 * ee / 3, "located_rows": eb.total,
                "count_kernel_ms": eix.kernel_time("count")[0], "locate_kernel_ms": eix.kernel_time("locate")[0],
                "parity": "tests/test_gpu_parity.py (oracle, all modes); profiles/ holds the run with the reference timed beside it"}
            # its own roofline block: the compulsory lines of THIS batch on THIS index (traced twins of the kernels) and, unless
            # --pmc off, the memory-side traffic from live rocprofv3 --pmc passes over a child run of the same workload
            e_cnt, e_n = eix.kernel_time("count")
            e_loc, _ = eix.kernel_time("locate")
            e_roof, e_kname, e_kms, e_comp, _ = roofline_block(eix, eix.rank_mode in (3, 4), eb, npats, ep, args.max_occs, e_cnt, e_loc, e_n)
            e_info = eix.pack_info()
            del eb
            eix.close()
            eix = None
            if args.pmc != "off" and world == 1:
                try:
                    e_args = argparse.Namespace(**vars(args))
                    e_args.workload = "eng"
                    tr, trs = pmc_traffic(e_args, e_kname, e_info)
                    add_traffic(e_roof, tr, trs, e_kms, e_comp)
                except Exception as ex:      # noqa: BLE001
                    log("cfg3 pmc pass failed:", repr(ex))
            extra["cfg3_text96_count_locate"]["roofline"] = e_roof
            extra["cfg3_text96_count_locate"]["index"] = e_info
        except Exception as ex:      # noqa: BLE001
            extra["cfg3_text96_count_locate"] = {"error": repr(ex)}

    if cpu:
        cpu["cgroup_cpu_quota"] = cpu_quota()       # CPUs this container may use on average (None: no quota); `cores` above is what was used
        cpu["gpu_vs_cpu"] = value / cpu["value"]     # a baseline, not a quality measure: the roofline fraction is
    wl = {"acgt": f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_rand 20-mers per GPU, count()+locate(max_occs={args.max_occs})",
          "acgt_hit": f"T_acgt(2^{args.text_log2}) femto index (default params), {npats} P_hit 20-mers per GPU, count()+locate(max_occs={args.max_occs})",
          "eng": f"T_eng(2^{args.text_log2}) femto index (default params), {npats} P_hit lengths 8..64 per GPU, count()+locate(max_occs={args.max_occs})"}[args.workload]
    out = {
        "metric": "patterns/sec (count+locate) on " + ("1 GiB index" if args.text_log2 == 30 else f"2^{args.text_log2} B index"),
        "value": value, "unit": "patterns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": wl, "text_bytes": n_text, "patterns_per_gpu": npats, "pattern_len": args.plen, "seed": args.seed,
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

"""Roofline accounting of bench.py: the byte models (distinct lines / line reads / SURVEY 8(d)'s formula on femto's own
operation counts) and the live rocprofv3 --pmc passes.  DESIGN.md section 4 explains the three byte counts."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

from .common import BENCH_PY, HBM_PEAK_GBS, ROOT, log, source_hash

KERNEL_NAMES = {(4, True): "femto_amd::count_direct_kernel<femto_amd::Pack2Policy, true", (3, True): "femto_amd::count_direct_kernel<femto_amd::PackPolicy, true",
                (1, False): "femto_amd::count_kernel_lane", (0, False): "femto_amd::count_kernel<32>"}
LOCATE_NAMES = {(4, True): "femto_amd::locate_walk_kernel<femto_amd::Pack2Policy>", (3, True): "femto_amd::locate_walk_kernel<femto_amd::PackPolicy>",
                (1, False): "femto_amd::locate_kernel_lane", (0, False): "femto_amd::locate_kernel<32>"}


def kernel_names(ix, direct):
    """names (prefixes) of the kernels the count and the locate timers bracket for this handle"""
    pi = ix.pack_info()
    cn, ln = KERNEL_NAMES[(ix.rank_mode, direct)], LOCATE_NAMES[(ix.rank_mode, direct)]
    if direct and ix.rank_mode == 3 and pi.get("rank_units"):
        cn = "femto_amd::count_direct_kernel<femto_amd::RumPolicy, true" if pi.get("rank_units_marked") else "femto_amd::count_direct_kernel<femto_amd::RuPolicy, true"
    if direct and ix.rank_mode == 4 and pi.get("char_rank_lines"):
        cn = "femto_amd::count_direct_kernel<femto_amd::IndPolicy, true"
    if direct:      # locate is fused into the row expansion: offsets from the resident suffix array (1) or by a walk per row (2)
        ln = "femto_amd::plan_rows_kernel<1," if pi.get("sa_full") else "femto_amd::plan_rows_kernel<2,"
    return cn, ln


PMC_REGEX = "count_direct_kernel|locate_walk_kernel|count_kernel|locate_kernel|count_tail_kernel|plan_rows_kernel"


def pmc_traffic(args, kname, pack_info, open_opts="", child_env=None):
    """HBM-side bytes per launch of kernel `kname`, measured NOW: two separate `rocprofv3 --pmc` passes over a short child
    run of this script (FETCH_SIZE; WRITE_SIZE + request counters -- never combined with any trace domain).  Per the
    guide (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes -> x2; both are KiB."""
    import csv
    import glob
    import shutil
    if not shutil.which("rocprofv3"):
        return None, None
    base = [sys.executable, BENCH_PY, "--pmc-child", "--steps", "2", "--warmup", "1", "--text-log2", str(args.text_log2),
            "--npats", str(args.npats), "--plen", str(args.plen), "--seed", str(args.seed), "--max-occs", str(args.max_occs),
            "--workload", args.workload, "--workdir", args.workdir] + (["--len-range", args.len_range] if args.len_range else []) + (
            ["--row-free"] if getattr(args, "row_free", False) else [])
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
        # optional third pass: a DRAM-only read counter, if this rocprofv3 / gfx950 exposes one (FETCH_SIZE and TCC_EA0_RDREQ count
        # requests the Infinity Cache serves as well, so they are memory-SIDE traffic, not HBM traffic)
        for ctr in ("TCC_EA0_RDREQ_DRAM_sum",):
            try:
                out = os.path.join(td, "pd")
                cmd = ["rocprofv3", "--pmc", ctr, "--kernel-include-regex", PMC_REGEX, "-f", "csv", "-d", out, "-o", "pmc", "--"] + base
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, cwd="/tmp", env=env)
                for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        for r in csv.DictReader(fh):
                            if _same_kernel(kname, r.get("Kernel_Name", "")):
                                means.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            except Exception:      # noqa: BLE001
                pass
    keys = ("level_table", "ktab_syms", "sa_full", "isa_full", "char_rank_lines", "context_table", "context_syms", "context2_syms", "rank_units")
    if child_info is None or any(child_info.get(k) != pack_info.get(k) for k in keys):
        log("pmc child built different structures, traffic not used:", child_info)
        return None, None
    if "FETCH_SIZE" not in means or "WRITE_SIZE" not in means:
        return None, None
    m = {k: sum(v) / len(v) for k, v in means.items()}
    traffic = 2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    dram = m.get("TCC_EA0_RDREQ_DRAM_sum")
    src = {"how": "live: 2 separate rocprofv3 --pmc passes over a 2-step child run of this script in this very run",
           "what": "memory-side requests of the L2s INCLUDING those the 256 MiB Infinity Cache serves (the guide: FETCH_SIZE derives from TCC_EA0_RDREQ, "
                   "Infinity-Cache hits are counted) -- an upper bound on HBM traffic, not HBM traffic; ~6.3 TB/s is what HBM itself streams",
           "dram_read_requests": dram, "dram_read_bytes_if_128B": (dram * 128.0) if dram else None,
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


def reference_format_block(cd, csub, npats, k_ms):
    """SURVEY 8(d)'s byte formula on the REFERENCE's own operation counts for this batch (oracle counters on `csub` patterns):
    N_rank x (12 + 64) + S bytes consumed + N_occ x 20 + N_mark x 8 -- what femto's algorithm would move on femto's format."""
    b = (cd["n_rank"] * (12 + 64) + cd["s_bytes"] + cd["n_occ"] * 20 + cd["n_mark"] * 8) / csub
    gbs = b * npats / (k_ms * 1e-3) / 1e9 if k_ms else None
    return {"bytes_per_pattern": b, "GBs": gbs, "x_peak": (gbs / HBM_PEAK_GBS) if gbs else None,
            "formula": f"SURVEY 8(d): N_rank*(12+64) + S_bytes + N_occ*20 + N_mark*8, oracle counters on {csub} patterns"}

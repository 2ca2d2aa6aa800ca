"""The secondary measurements of bench.py's N = 1 run (each an `extra` JSON line of its own before the headline line; a failure
in one of them never costs the headline).  Every function returns a dict."""
import json
import os
import subprocess
import tempfile
import time

import numpy as np

from .batch import Batch, row_free_steps, timed_steps
from .common import HBM_PEAK_GBS, cpu_quota, log
from .roofline import add_traffic, pmc_traffic, reference_format_block, roofline_block


class Ctx:
    """what the extras share with main(): args, torch / femto_amd / textgen modules, device, stream, paths, sizes"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def budget_extra(c, batch, plen, ref_results, budget="4x"):
    """A footprint-bounded open of the SAME index on the headline batch (random 20-mers) and on 20-mers sampled from the text,
    with its own roofline block and live PMC traffic.  budget = "4x": hbm_budget_bytes = 4 x text bytes (round-3 verdict,
    task 1): packed lines, rank units, sampled marks and the level table the rest pays for; no dense suffix arrays, no text.
    budget = "default": femto_amd_open as a drop-in caller gets it -- the library's default bound, min(free / 4, max(8 x text, 2 GiB))."""
    args, torch, femto_amd, tg, dev, local_rank, stream = c.args, c.torch, c.femto_amd, c.tg, c.dev, c.local_rank, c.stream
    index_path, text_path, n_text = c.index_path, c.text_path, c.n_text
    npats = args.npats
    if budget == "default":
        bix = femto_amd.Index(index_path, device=local_rank)
        held = bix.structures()
        budget_bytes = held["hbm_budget"]
        out = {"what": f"same index opened with plain femto_amd_open: the library's DEFAULT bound (hbm_budget_bytes auto = {budget_bytes} bytes here)",
               "structures": held, "index": bix.pack_info()}
        assert held["hbm_budget_is_default"] == 1 and held["hbm_allocated"] <= budget_bytes
        pmc_opts = ""
    else:
        budget_bytes = 4 * n_text
        bix = femto_amd.Index(index_path, device=local_rank, options={"hbm_budget_bytes": budget_bytes})
        out = {"what": f"same index opened with femto_amd_open_opts(hbm_budget_bytes = 4 x text = {budget_bytes}): search steps on the rank units / packed lines, "
                       "no dense suffix arrays, no text tail", "structures": bix.structures(), "index": bix.pack_info()}
        pmc_opts = f"hbm_budget_bytes={budget_bytes}"
    try:
        steps = max(5, min(args.steps, 100))
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
        if args.pmc != "off" and budget != "default":      # (the default bound's counter passes: tools/profile_round.sh --open-opts hbm_budget_bytes=-1; two more child runs would cost the default bench run ~45 s)
            try:
                tr, trs = pmc_traffic(args, kname, info, open_opts=pmc_opts or "hbm_budget_bytes=-1")
                add_traffic(roof, tr, trs, k_ms, comp)
            except Exception as ex:      # noqa: BLE001
                log("budget pmc pass failed:", repr(ex))
    except Exception as ex:      # noqa: BLE001
        out["error"] = repr(ex)
    finally:
        if bix is not None:
            bix.close()
    return out


def mode1_extra(c, ix, batch, ref_results, cd_count, csub):
    """SURVEY 8(d)'s own kernel family: femto's wavelet tree itself (mode 1: one lane per pattern on the derived segment lines
    of femto's RLE / literal sequences, batch ordered by suffix).  The only family 8(d)'s byte formula describes."""
    args, torch, stream = c.args, c.torch, c.stream
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


def shim_locate_sampled_extra(args, index_path, plen, flat, located_rows):
    """parallel_locate through the shim on a batch where EVERY pattern matches (20-mers sampled from the text): the callee
    malloc()s offsets[i] once per pattern (femto.c:372-386 -- the contract: the caller free()s each), 10 M times per call"""
    from oracle import pyoracle as po
    if not os.path.exists(po.REF_TOOL_AMD):
        return {"error": "oracle/_ref/ref_tool_amd not built"}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        pf = os.path.join(td, "p.fpat")
        po.write_fpat_flat(pf, plen, flat)
        reps = 4
        o_ = subprocess.run([po.REF_TOOL_AMD, "bench", index_path, pf, "locate", str(args.max_occs), "1", str(reps)],
                            check=True, stdout=subprocess.PIPE, timeout=600).stdout.decode()
        tj = json.loads(o_.strip().splitlines()[-1])
        return {"what": "parallel_locate through integration/femto_amd_shim.c on 20-mers SAMPLED from the text: every pattern matches, so the "
                        f"callee malloc()s one offsets[i] per pattern ({len(plen)} per call, free()d by the caller between calls, outside the timed call); "
                        f"1 warm-up + {reps} timed calls; value = mean",
                "value": len(plen) / tj["mean_s"], "best": len(plen) / tj["best_s"], "unit": "patterns/s", "ms": 1e3 * tj["mean_s"], "best_ms": 1e3 * tj["best_s"],
                "results": int(tj["results"]), "equal_to_device_path": bool(int(tj["results"]) == int(located_rows))}


def p_hit_extra(c, ix, hb):
    """every pattern occurs and is located (same index, P_hit 20-mers)"""
    args, torch, stream = c.args, c.torch, c.stream
    steps = max(3, min(args.steps, 50))
    el, (c_ms, c_n), (l_ms, _) = timed_steps(torch, ix, hb, args.max_occs, stream, steps)
    out = {"workload": f"{hb.n} 20-mers sampled from the text, count()+locate(max_occs={args.max_occs})",
           "value": hb.n * steps / el, "unit": "patterns/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
           "located_rows": hb.total, "count_kernel_ms": c_ms, "locate_kernel_ms": l_ms}
    try:
        roof, _, _, _, _ = roofline_block(ix, True, hb, hb.n, hb.plen, args.max_occs, c_ms, l_ms, c_n)
        out["roofline"] = {k: roof[k] for k in ("achieved", "frac", "kernel", "kernel_ms", "compulsory_bytes_per_launch", "line_reads")}
    except Exception as ex:      # noqa: BLE001
        out["roofline"] = {"error": repr(ex)}
    out["row_free"] = row_free_steps(torch, ix, hb, args.max_occs, stream, steps)
    try:
        ix.set_option("trace_row_free", 1)
        rr, _, _, _, _ = roofline_block(ix, True, hb, hb.n, hb.plen, args.max_occs, out["row_free"]["count_kernel_ms"], out["row_free"]["locate_kernel_ms"], c_n)
        out["row_free"]["roofline"] = {k: rr[k] for k in ("achieved", "frac", "kernel", "kernel_ms", "compulsory_bytes_per_launch", "line_reads")}
    except Exception as ex:      # noqa: BLE001
        out["row_free"]["roofline"] = {"error": repr(ex)}
    finally:
        ix.set_option("trace_row_free", 0)
    return out


def keys_extra(c, ix, batch, ref_results):
    """The headline batch once more as 64-bit KEYS (femto_amd_pack_keys_device: 3 bits per DNA symbol, packed once, untimed --
    the form a caller that keeps its patterns in HBM would hold them in) with the ranges returned as int32 pairs:
    28 instead of 80 bytes streamed per pattern around the same search.  Same results, checked; an `extra`
    line, never the headline (whose input is the reference's own alpha_t symbols)."""
    args, torch, dev, stream = c.args, c.torch, c.dev, c.stream
    npats = batch.n
    first, last, g_noccs, g_ost, g_offs = ref_results
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
    ksteps = max(3, min(args.steps, 100))
    t0 = time.perf_counter()
    for _ in range(ksteps):
        kstep()
    torch.cuda.synchronize()
    ke = time.perf_counter() - t0
    ix.kernel_time_enable(False)
    pairs = k_r32.cpu().numpy().reshape(npats, 2)
    same = bool(np.array_equal(pairs[:, 0], first) and np.array_equal(pairs[:, 1], last) and np.array_equal(k_noccs.cpu().numpy(), g_noccs)
                and np.array_equal(k_ost.cpu().numpy(), g_ost) and np.array_equal(k_offs[:len(g_offs)].cpu().numpy(), g_offs))
    assert same, "the key path's results differ from the symbol path's"
    return {"what": "the headline batch as 64-bit keys in, int32 (first,last) pairs + row counts + located offsets out (femto_amd_locate_keys_device)",
            "value": npats * ksteps / ke, "unit": "patterns/s", "ms_per_step": 1e3 * ke / ksteps, "steps": ksteps,
            "count_kernel_ms": ix.kernel_time("count")[0], "streamed_bytes_per_pattern": 28, "equal_to_symbol_path": same}


def dna_motif(rng, k):
    """a random DNA motif of k terms: bases, two-base classes, two-way alternations of dimers, optional bases"""
    out = b""
    for _ in range(k):
        r = rng.random()
        if r < 0.70:
            out += bytes([b"ACGT"[rng.integers(0, 4)]])
        elif r < 0.85:
            out += b"[" + bytes(sorted(set(b"ACGT"[i] for i in rng.integers(0, 4, 2)))) + b"]"
        elif r < 0.93:
            out += (b"(" + bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2)) + b"|" + bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2)) + b")")
        else:
            out += bytes([b"ACGT"[rng.integers(0, 4)]]) + b"?"
    return out


def regexp_workloads(femto_amd, seed, n_exact=20000, n_approx=20000):
    """the two automaton batches of profiles/r03_regexp_batch.txt: exact motifs of 14-18 terms, APPROX 1 motifs of 16-20 terms"""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    exact = [femto_amd.Nfa.from_regex(dna_motif(rng, int(rng.integers(14, 19)))) for _ in range(n_exact)]
    approx = [femto_amd.Nfa.from_regex(dna_motif(rng, int(rng.integers(16, 21))), (1, 1, 1, 1)) for _ in range(n_approx)]
    return {"exact_motifs_14_18": exact, "approx1_motifs_16_20": approx}


def regexp_extra(c, ix, n_exact=20000, n_approx=20000, ref_exact=200, ref_approx=40):
    """Automaton search (do_regexp_query for a BATCH of automata, femto_amd_nfa_search_batch) on the headline
    index: both batches timed (wall of the call and the search kernels' own time from HIP events), the genuine reference --
    one thread, its only mode; the time inside femto_run_query, process start and index open excluded -- on the first
    automata of each batch, result lists compared bit for bit."""
    from oracle import pyoracle as po
    femto_amd = c.femto_amd
    out = {"what": "femto_amd_nfa_search_batch on the headline index: one call per batch (automata compiled and marshalled into femto_amd_nfa_t beforehand; flattening, upload, search and result sort inside the timed call)"}
    work = regexp_workloads(femto_amd, c.args.seed, n_exact, n_approx)
    ix.nfa_search_batch(work["exact_motifs_14_18"][:128], max_results=1 << 22)      # warm-up: kernel load, arena allocation
    for name, nfas in work.items():
        pre = femto_amd.NfaBatch(nfas)        # the C caller's array of femto_amd_nfa_t, built outside the timed call
        ix.kernel_time_reset()
        ix.kernel_time_enable(True)
        t0 = time.perf_counter()
        r_start, r_first, r_last, r_len, r_cost, r_status = ix.nfa_search_batch(pre, max_results=1 << 25)
        dt = time.perf_counter() - t0
        ix.kernel_time_enable(False)
        k_ms, k_n = ix.kernel_time("regexp")
        e = {"automata": len(nfas), "nodes_avg": float(np.mean([a.num_nodes for a in nfas])), "value": len(nfas) / dt, "unit": "automata/s", "ms": 1e3 * dt,
             "kernel_ms_total": k_ms * k_n, "kernel_launches": k_n, "kernel_automata_per_s": len(nfas) / (k_ms * k_n * 1e-3) if k_n else None,
             "result_ranges": int(len(r_first)), "not_ok": int((r_status != 0).sum())}
        if po.have_ref():
            m = ref_exact if name.startswith("exact") else ref_approx
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                ref = po.ref_regexp_nfa(c.index_path, nfas[:m], td, timeout=900)
            same = all(rr[0] == int(r_status[i]) and np.array_equal(rr[1], r_first[r_start[i]:r_start[i + 1]])
                       and np.array_equal(rr[2], r_last[r_start[i]:r_start[i + 1]]) and np.array_equal(rr[3], r_len[r_start[i]:r_start[i + 1]])
                       and np.array_equal(rr[4], r_cost[r_start[i]:r_start[i + 1]]) for i, rr in enumerate(ref))
            assert same, f"automaton search ({name}): GPU result lists differ from the genuine do_regexp_query"
            qs = po.last_regexp_query_s
            e["cpu_baseline"] = {"value": (m / qs) if qs else None, "unit": "automata/s", "kind": "reference", "cores": 1,
                                 "sample": f"the first {m} automata through setup_regexp_query_take_nfa + do_regexp_query (ref_tool regexp_nfa); time inside "
                                           "femto_run_query only, index in page cache", "bit_exact_vs_gpu": True}
            if qs:
                e["gpu_vs_cpu"] = e["value"] / (m / qs)
        out[name] = e
    ex = out["exact_motifs_14_18"]
    out.update({"value": ex["value"], "unit": "automata/s", "ms": ex["ms"]})
    return out


def host_pointer_extra(c, ix, batch, plen, flat, first, last):
    """PCIe-inclusive rate of the host-pointer entry point (patterns and results in pageable host memory):
    never the headline value, reported for the drop-in caller's benefit"""
    femto_amd = c.femto_amd
    npats = batch.n
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
    return {"what": "femto_amd_count_flat on the same 10M-pattern batch, pageable host arrays in and out (staging threads + PCIe + kernels, pipelined in "
                    "1M-pattern stages, three in flight); value = fastest of 5 calls after a warm-up (single calls run longer once the spinning "
                    "staging threads have spent the container's CPU quota -- cgroup_cpu_quota CPUs on average: mean_ms)",
            "value": npats / hs, "unit": "patterns/s", "ms": 1e3 * hs, "mean_ms": 1e3 * sum(hts[1:]) / len(hts[1:]),
            "host_hardware_threads": os.cpu_count(), "cgroup_cpu_quota": cpu_quota(),
            "equal_to_device_path": bool(np.array_equal(hf_, first) and np.array_equal(hl_, last))}


def mode0_extra(c, ix, batch, ref_results):
    """BASELINE.json north_star's literal kernel: one WAVEFRONT per query walking femto's own group tables and varbyte sums with
    __ballot / ds_bpermute, no derived tables (rank mode 0, kernels.hip.hpp).  Measured once per round next to mode 1 so that
    DESIGN.md's reason for not shipping it as the default rests on a current number."""
    args, torch, stream = c.args, c.torch, c.stream
    out = {"what": "the headline batch through rank mode 0 (count_kernel<32>: wavefront-cooperative walk of femto's raw A/S/D tables)"}
    old = ix.rank_mode
    try:
        ix.set_rank_mode(0)
        el, (c_ms, c_n), (l_ms, _) = timed_steps(torch, ix, batch, args.max_occs, stream, 2, warm=1)
        first, last, noccs, ost, offs = ref_results
        same = bool(np.array_equal(batch.d_res[0].cpu().numpy(), first) and np.array_equal(batch.d_res[1].cpu().numpy(), last)
                    and np.array_equal(batch.offsets[:batch.total].cpu().numpy(), offs))
        assert same, "mode 0: results differ from the packed path's"
        out.update({"value": batch.n * 2 / el, "unit": "patterns/s", "ms_per_step": 1e3 * el / 2, "count_kernel_ms": c_ms, "locate_kernel_ms": l_ms,
                    "equal_to_headline_results": same, "kernel": "femto_amd::count_kernel<32>"})
    except Exception as ex:      # noqa: BLE001
        out["error"] = repr(ex)
    finally:
        ix.set_rank_mode(old)
    return out


def cfg3_extra(c, world):
    """BASELINE configs[2]: a sigma~96 text of the same size, sampled patterns of length 8..64 (the two-level 16-ary lines +
    per-character rank lines + hashed context tables, mode 4), with its own roofline block (traced twins + live PMC), a
    bit-check of the first 100 k patterns of the timed batch against the oracle (count and locate) and the genuine
    reference (count), and the located rows resolved to (document, offset) on the device."""
    args, torch, femto_amd, tg, dev, local_rank, stream = c.args, c.torch, c.femto_amd, c.tg, c.dev, c.local_rank, c.stream
    npats, n_text = args.npats, c.n_text
    e_path = os.path.join(args.workdir, f"eng_2p{args.text_log2}_s{args.seed}")
    e_text = tg.t_eng_torch(n_text, args.seed, f"cuda:{local_rank}")
    if not os.path.exists(os.path.join(e_path, "_femto_index")):
        femto_amd.build_index(e_path, [e_text], params=None, infos=["bench"], device=local_rank)
    eix = femto_amd.Index(e_path, device=local_rank, options={"hbm_budget_bytes": femto_amd.BUDGET_ALL})
    try:
        ep, ef = tg.p_hit(8, 64, npats, args.seed + 3000, e_text)
        del e_text
        eb = Batch(torch, dev, ep, ef)
        e_steps = max(3, min(args.steps, 20))
        ee, (e_cnt, e_n), (e_loc, _) = timed_steps(torch, eix, eb, args.max_occs, stream, e_steps)
        out = {"workload": f"T_eng(2^{args.text_log2}) sigma~96 index, {npats} sampled patterns of length 8..64, count()+locate(max_occs={args.max_occs})",
               "rank_mode": {4: "pack2", 3: "pack", 1: "lane", 0: "raw"}[eix.rank_mode],
               "value": npats * e_steps / ee, "unit": "patterns/s", "ms_per_step": 1e3 * ee / e_steps, "steps": e_steps, "located_rows": eb.total,
               "count_kernel_ms": e_cnt, "locate_kernel_ms": e_loc}
        # ---- bit-check inside the bench, like the headline's: the first 100 k patterns of the timed batch
        try:
            from oracle import pyoracle as po
            m = min(100_000, npats)
            g_first, g_last = eb.d_res[0][:m].cpu().numpy(), eb.d_res[1][:m].cpu().numpy()
            g_noccs = eb.d_noccs[:m].cpu().numpy()
            g_rows = int(eb.d_ostarts[m].item())
            g_offs = eb.offsets[:g_rows].cpu().numpy()
            o = po.Oracle(e_path)
            s_flat = ef[:int(eb.starts[m - 1] + ep[m - 1])]
            nthr = min(64, os.cpu_count() or 1)
            of, ol = o.count_flat(ep[:m], s_flat, eb.starts[:m], threads=nthr)
            on, oo = o.locate_flat(ep[:m], s_flat, eb.starts[:m], args.max_occs, threads=nthr)
            o.close()
            assert np.array_equal(of, g_first) and np.array_equal(ol, g_last), "cfg 3: GPU count differs from the oracle"
            assert np.array_equal(on, g_noccs) and np.array_equal(oo, g_offs), "cfg 3: GPU locate differs from the oracle"
            chk = {"patterns": m, "rows": g_rows, "oracle_count_locate": True}
            if po.have_ref():
                sub = min(m, 20_000)
                with tempfile.TemporaryDirectory() as td:
                    pf, rf = os.path.join(td, "p.fpat"), os.path.join(td, "r.bin")
                    po.write_fpat_flat(pf, ep[:sub], ef[:int(eb.starts[sub - 1] + ep[sub - 1])])
                    subprocess.run([po.REF_TOOL, "count", e_path, pf, rf], check=True, stdout=subprocess.PIPE, timeout=600)
                    ref = np.fromfile(rf, dtype=np.int64)
                assert np.array_equal(ref[:sub], g_first[:sub]) and np.array_equal(ref[sub:], g_last[:sub]), "cfg 3: GPU ranges differ from the genuine reference"
                chk["reference_parallel_count"] = sub
            out["bit_exact"] = chk
        except AssertionError:
            raise
        except Exception as ex:      # noqa: BLE001
            out["bit_exact"] = {"error": repr(ex)}
        # ---- the step right after locate: every located row's (document, offset) on the device, in the same stream
        try:
            rows = eb.total
            d32 = torch.empty(max(rows, 1), dtype=torch.int32, device=dev)
            doff = torch.empty(max(rows, 1), dtype=torch.int64, device=dev)
            for _ in range(2):
                eix.resolve_device(eb.offsets.data_ptr(), rows, d_doc32=d32.data_ptr(), d_doc_offset=doff.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            eix.kernel_time_reset()
            eix.kernel_time_enable(True)
            for _ in range(5):
                eix.resolve_device(eb.offsets.data_ptr(), rows, d_doc32=d32.data_ptr(), d_doc_offset=doff.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            eix.kernel_time_enable(False)
            r_ms, r_n = eix.kernel_time("resolve")
            ok = bool((d32 == 0).all().item() and torch.equal(doff, eb.offsets[:rows]))      # one document: offset in document = offset
            out["resolve_device"] = {"what": "femto_amd_resolve_device on the located rows of one step (resolve_location, index.c:1587): int32 document + int64 offset in document",
                                     "rows": rows, "kernel_ms": r_ms, "value": rows / (r_ms * 1e-3) if r_ms else None, "unit": "offsets/s",
                                     "GBs": rows * 20 / (r_ms * 1e-3) / 1e9 if r_ms else None, "frac_of_hbm_peak": rows * 20 / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if r_ms else None,
                                     "bytes_model": "8 B in + 12 B out per offset, streamed", "correct": ok}
            assert ok, "cfg 3: device resolve of a one-document index must return document 0 and the offset itself"
            del d32, doff
        except AssertionError:
            raise
        except Exception as ex:      # noqa: BLE001
            out["resolve_device"] = {"error": repr(ex)}
        # its own roofline block: the compulsory lines of THIS batch on THIS index (traced twins of the kernels) and, unless
        # --pmc off, the memory-side traffic from live rocprofv3 --pmc passes over a child run of the same workload
        e_roof, e_kname, e_kms, e_comp, _ = roofline_block(eix, eix.rank_mode in (3, 4), eb, npats, ep, args.max_occs, e_cnt, e_loc, e_n)
        e_info = eix.pack_info()
        out["structures"] = eix.structures()
        # ---- the same batch through the row-free form (what parallel_locate returns: noccs + offsets), results compared
        try:
            eb.step(eix, args.max_occs, stream)
            torch.cuda.synchronize()
            eb.total = int(eb.d_total[0].item())
            rf = row_free_steps(torch, eix, eb, args.max_occs, stream, e_steps)
            eix.set_option("trace_row_free", 1)
            try:
                rr, _, _, rcomp, _ = roofline_block(eix, eix.rank_mode in (3, 4), eb, npats, ep, args.max_occs, rf["count_kernel_ms"], rf["locate_kernel_ms"], e_n)
            finally:
                eix.set_option("trace_row_free", 0)
            rf["roofline"] = {k: rr[k] for k in ("achieved", "frac", "kernel", "kernel_ms", "compulsory_bytes_per_launch", "line_reads")}
            out["row_free"] = rf
        except AssertionError:
            raise
        except Exception as ex:      # noqa: BLE001
            out["row_free"] = {"error": repr(ex)}
        eix.close()
        eix = None
        # ---- the same index as a drop-in opens it (plain femto_amd_open: the library's default bound), with rows and row-free
        try:
            dix = femto_amd.Index(e_path, device=local_rank)
            try:
                d_steps = max(3, min(args.steps, 5))
                de, (d_cnt, _), (d_loc, _) = timed_steps(torch, dix, eb, args.max_occs, stream, d_steps)
                d = {"what": "the same index and batch on a handle opened with plain femto_amd_open (the default bound)", "value": npats * d_steps / de, "unit": "patterns/s",
                     "ms_per_step": 1e3 * de / d_steps, "count_kernel_ms": d_cnt, "locate_kernel_ms": d_loc, "structures": dix.structures(), "index": dix.pack_info()}
                d["row_free"] = row_free_steps(torch, dix, eb, args.max_occs, stream, d_steps)
                out["default_open"] = d
            finally:
                dix.close()
        except AssertionError:
            raise
        except Exception as ex:      # noqa: BLE001
            out["default_open"] = {"error": repr(ex)}
        del eb
    finally:
        if eix is not None:
            eix.close()
    if args.pmc != "off" and world == 1:
        try:
            import argparse
            e_args = argparse.Namespace(**vars(args))
            e_args.workload = "eng"
            tr, trs = pmc_traffic(e_args, e_kname, e_info)
            add_traffic(e_roof, tr, trs, e_kms, e_comp)
        except Exception as ex:      # noqa: BLE001
            log("cfg3 pmc pass failed:", repr(ex))
    out["roofline"] = e_roof
    out["index"] = e_info
    return out

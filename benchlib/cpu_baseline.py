"""The CPU path timed beside the GPU on a bounded sample of the same batch (rank 0, N = 1 contract: ~10-30 s of CPU work):
the genuine reference (oracle/_ref, kind "reference") when it was built, else the oracle's restatement (kind "port") -- and the
bit-exact check of the GPU's results against both.  Only bench.py's cpu_baseline leg (and tests/) may touch oracle/."""
import json
import os
import subprocess
import tempfile
import time

import numpy as np


def cpu_baseline(args, index_path, plen, starts, flat, first, last, g_noccs, g_ost, g_offs, value):
    """returns (cpu_baseline dict, reference_equivalent_work dict, count counters, counter sample size)"""
    from oracle import pyoracle as po
    npats = len(plen)
    host_cores = os.cpu_count() or 1
    sample = min(args.cpu_sample, npats)
    if sample <= 0:
        return None, None, None, 0
    o = po.Oracle(index_path)
    s_plen, s_starts = plen[:sample], starts[:sample]
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
    return cpu, ref_work, cd_count, csub

#!/usr/bin/env python3
"""tools/promote_profiles.py <tag>...: copy what tools/profile_round.sh wrote under gpurun_out/<tag>/ into profiles/
(bench line + rocprofv3 kernel stats + per-kernel PMC means) and refresh profiles/latest_pmc.json from r02_default."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import common as bench  # noqa: E402  (source_hash)

KEEP = ("count_direct_kernel", "count_tail_kernel", "plan_rows_kernel", "plan_super_kernel", "locate_walk_kernel", "count_keys_kernel",
        "count_kernel", "locate_kernel")


def main():
    for tag in sys.argv[1:]:
        src = os.path.join(ROOT, "gpurun_out", tag)
        lines = [ln for ln in open(os.path.join(src, "bench.json")).read().strip().splitlines() if ln.startswith("{")]
        line = lines[-1]      # the headline; the `extra` lines before it are kept too (one JSON object per line)
        for ln in lines:
            json.loads(ln)
        open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w").write("\n".join(lines) + "\n")
        rec = json.loads(line)
        sh = ((rec.get("roofline") or {}).get("traffic_source") or {}).get("source_hash") or bench.source_hash()
        out = [f"# {tag}: tools/profile_round.sh (rocprofv3 --kernel-trace --stats of `python bench.py ...`, then separate --pmc passes)",
               f"# source hash of femto_amd/csrc at measurement time: {sh}"]
        keep = False
        for ln in open(os.path.join(src, "summary.txt")):
            ln = ln.rstrip("\n")
            if ln.startswith("== "):
                out.append(ln)
                keep = False
                continue
            if ln.startswith("{'Name'"):
                if "_traced::" in ln:
                    continue
                out.append(ln[:420])
                continue
            if ln and not ln.startswith(" "):      # a kernel header in the PMC section
                keep = any(k in ln for k in KEEP) and "_traced" not in ln
            if keep:
                out.append(ln)
        open(os.path.join(ROOT, "profiles", f"{tag}_stats.txt"), "w").write("\n".join(out) + "\n")
        if tag.split("_", 1)[-1] == "default":      # (rNN_default only: not rNN_eng_default)
            pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
            d = json.loads(line)
            kname = d["roofline"]["kernel"]
            for k, v in pmc.items():
                if "count_direct_kernel" in k and "_traced" not in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    rec = {"npats": d["config"]["patterns_per_gpu"], "text_log2": 30, "workload": "acgt", "kernel": kname,
                           "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                           "TCC_EA0_RDREQ": v.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_128B": v.get("TCC_EA0_RDREQ_128B_sum"),
                           "hbm_bytes_per_launch": 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024,
                           "source_hash": ((d.get("roofline") or {}).get("traffic_source") or {}).get("source_hash") or bench.source_hash(),
                           "note": "fallback only: bench.py measures traffic live (two rocprofv3 --pmc passes in the same run) and accepts this "
                                   "file only when source_hash matches the kernel sources; 2 x FETCH_SIZE KiB (gfx950 tallies 128-B requests "
                                   "as 64 B) + WRITE_SIZE KiB", "source": f"profiles/{tag}_stats.txt"}
                    json.dump(rec, open(os.path.join(ROOT, "profiles", "latest_pmc.json"), "w"), indent=1)
        print("promoted", tag)


if __name__ == "__main__":
    main()

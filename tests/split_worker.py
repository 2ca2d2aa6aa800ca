"""Worker of test_range_split_across_processes: one rank of a 2-process range-split on one GPU.
argv: index_dir golden.npz out_dir.  Writes out_dir/ok<rank> when every check passed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import femto_amd  # noqa: E402
from femto_amd import parallel  # noqa: E402


def main():
    index, gold_path, out_dir = sys.argv[1:4]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    g = np.load(gold_path)
    ix = parallel.open_range_split(index, dev)
    info = ix.split_info()
    assert info["nparts"] == 2 and info["part"] == rank and info["seg_bytes"] > 0
    plen = g["pat_len"].astype(np.int32)
    flat = g["pat_flat"].astype(np.uint16)
    starts = np.zeros(len(plen), dtype=np.int64)
    starts[1:] = np.cumsum(plen[:-1])
    n = ix.info.total_length
    ch, occ, off = ix.block_requests(np.arange(n, dtype=np.int64))
    assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])
    first, last = ix.count_flat(plen, flat, starts)
    assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"])
    for k in g.files:
        if k.startswith("loc") and k.endswith("_noccs"):
            mo = int(k[3:-6])
            nocc, offs = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(nocc, g[k]) and np.array_equal(offs, g[f"loc{mo}_offs"])
    # sharded batch + gather, as the replicated path does
    res = parallel.sharded_count(ix.count_flat, plen, flat, starts)
    if rank == 0:
        assert np.array_equal(res[0], g["count_first"]) and np.array_equal(res[1], g["count_last"])
    dist.barrier()      # keep every owner's memory alive until all ranks are done reading it
    ix.close()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker of test_striped_index_across_processes: one rank of a 2-process striped index on one GPU.
argv: out_dir then (index_dir golden.npz) pairs.  Rank 0 derives the striped index and serves its stripes as file
descriptors; rank 1 attaches (femto_amd_open_striped_client).  Writes out_dir/ok<rank> when every check passed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import femto_amd  # noqa: E402
from femto_amd import parallel  # noqa: E402


def check(ix, g, want_mode):
    assert ix.rank_mode == want_mode, (ix.rank_mode, want_mode)
    plen = g["pat_len"].astype(np.int32)
    flat = g["pat_flat"].astype(np.uint16)
    starts = np.zeros(len(plen), dtype=np.int64)
    starts[1:] = np.cumsum(plen[:-1])
    n = ix.info.total_length
    for mode in (want_mode, 1):                      # the fast path of this alphabet and the wavelet path
        ix.set_rank_mode(mode)
        ch, occ, off = ix.block_requests(np.arange(n, dtype=np.int64))
        assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"])
        for k in g.files:
            if k.startswith("loc") and k.endswith("_noccs"):
                mo = int(k[3:-6])
                nocc, offs = ix.locate_flat(plen, flat, starts, mo)
                assert np.array_equal(nocc, g[k]) and np.array_equal(offs, g[f"loc{mo}_offs"])
    ix.set_rank_mode(want_mode)
    # the enqueue-only chain on device-resident inputs (what bench.py times)
    dev = torch.device("cuda", 0)
    d_plen, d_flat, d_starts = (torch.from_numpy(a).to(dev) for a in (plen, flat.view(np.int16), starts))
    m = len(plen)
    res = torch.empty((2, m), dtype=torch.int64, device=dev)
    noccs = torch.empty(m, dtype=torch.int32, device=dev)
    ost = torch.empty(m + 1, dtype=torch.int64, device=dev)
    want_n, want_o = g["loc7_noccs"], g["loc7_offs"]
    offs = torch.empty(len(want_o) + 8, dtype=torch.int64, device=dev)
    tot = torch.zeros(2, dtype=torch.int64, device=dev)
    ix.locate_device(m, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), 7, res[0].data_ptr(), res[1].data_ptr(),
                     noccs.data_ptr(), ost.data_ptr(), offs.data_ptr(), offs.numel(), tot.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(want_o) and int(tot[1]) == 0
    assert np.array_equal(noccs.cpu().numpy(), want_n) and np.array_equal(offs[:len(want_o)].cpu().numpy(), want_o)
    assert np.array_equal(res[0].cpu().numpy(), g["count_first"]) and np.array_equal(res[1].cpu().numpy(), g["count_last"])


def main():
    import faulthandler
    faulthandler.enable()
    out_dir = sys.argv[1]
    pairs = list(zip(sys.argv[2::3], sys.argv[3::3], sys.argv[4::3]))
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    for k, (index, gold_path, mode) in enumerate(pairs):
        g = np.load(gold_path)
        sock = os.path.join(out_dir, f"stripes{k}.sock")
        ix, keep = parallel.open_striped_shared(index, 0, sock, devices=[0, 0])
        pi = ix.pack_info()
        assert pi["level_table"] and pi["sa_full"], pi
        check(ix, g, int(mode))
        dist.barrier()
        ix.close()
        if keep is not None:
            keep.close()
        dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

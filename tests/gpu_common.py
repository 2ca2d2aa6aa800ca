"""helpers shared by the `-m gpu` test modules"""
import os

import pytest

import femto_amd

# 3: packed small-alphabet lines (default when the index has <= 8 characters); 4: two-level 16-ary lines (default for
# 9..256 characters); 1: lane per query on femto's wavelet tree (default otherwise); 0: wavefront-per-query raw walk
MODES = [3, 4, 1, 0]


def _torchrun(nproc, script_and_args, env, cwd=None, attempts=2):
    """python -m torch.distributed.run on 127.0.0.1 with a free port; one retry (a port can be taken between probing and use)"""
    import socket
    import subprocess
    import sys
    out = None
    for _ in range(attempts):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_and_args,
                             env=env, capture_output=True, text=True, timeout=600, cwd=cwd)
        if out.returncode == 0:
            break
    return out


def _open(path, mode=None):
    """open on GPU 0; mode 4 is built for small alphabets too (FEMTO_AMD_PACK2=1) so that every fixture exercises it"""
    old = os.environ.get("FEMTO_AMD_PACK2")
    os.environ["FEMTO_AMD_PACK2"] = "1"
    try:
        ix = femto_amd.Index(path, device=0)
    finally:
        if old is None:
            del os.environ["FEMTO_AMD_PACK2"]
        else:
            os.environ["FEMTO_AMD_PACK2"] = old
    if mode is not None:
        _set_mode(ix, mode)
    return ix


def _set_mode(ix, mode):
    if mode == 3 and not ix.pack_info()["available"]:
        ix.close()
        pytest.skip("more than 8 distinct characters: no packed lines for this index")
    if mode == 4 and not ix.pack_info()["available2"]:
        ix.close()
        pytest.skip("more than 256 distinct characters: no two-level lines for this index")
    ix.set_rank_mode(mode)
    assert ix.rank_mode == mode


def device_locate(ix, plen, flat, starts, max_occs, capacity, row_free=False):
    """femto_amd_locate_device (the one-call device chain: count -> plan_rows with the walk inside) on host arrays:
    (first, last, noccs, out_starts, offsets[:min(total, capacity)], total).  row_free: the form without row arrays
    (d_first = d_last = NULL: parallel_locate's own results); first / last come back as None."""
    import numpy as np
    import torch
    dev = "cuda:0"
    n = len(plen)
    d_plen, d_flat, d_starts = torch.from_numpy(plen).to(dev), torch.from_numpy(flat.view(np.int16)).to(dev), torch.from_numpy(starts).to(dev)
    f, l = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
    noccs = torch.zeros(n, dtype=torch.int32, device=dev)
    ostarts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offs = torch.full((capacity,), -7, dtype=torch.int64, device=dev)
    total = torch.zeros(2, dtype=torch.int64, device=dev)
    ix.locate_device(n, d_plen.data_ptr(), d_flat.data_ptr(), d_starts.data_ptr(), max_occs, 0 if row_free else f.data_ptr(), 0 if row_free else l.data_ptr(),
                     noccs.data_ptr(), ostarts.data_ptr(), offs.data_ptr(), capacity, total.data_ptr())
    torch.cuda.synchronize()
    tot = int(total[0])
    if row_free:
        assert not f.any() and not l.any()
        return (None, None, noccs.cpu().numpy(), ostarts.cpu().numpy(), offs[:min(tot, capacity)].cpu().numpy(), tot)
    return (f.cpu().numpy(), l.cpu().numpy(), noccs.cpu().numpy(), ostarts.cpu().numpy(), offs[:min(tot, capacity)].cpu().numpy(), tot)


def assert_row_free_equals(ix, plen, flat, starts, max_occs, noccs, offs, what=""):
    """the row-free form of the device chain returns the same noccs / out_starts / offsets / total as the form with rows"""
    import numpy as np
    _, _, dn, dst, do, dtot = device_locate(ix, plen, flat, starts, max_occs, len(offs) + 16, row_free=True)
    want_st = np.zeros(len(plen) + 1, dtype=np.int64)
    want_st[1:] = np.cumsum(noccs.astype(np.int64))
    assert dtot == len(offs), (what, dtot, len(offs))
    assert np.array_equal(dn, noccs), what
    assert np.array_equal(dst, want_st), what
    assert np.array_equal(do, offs), what

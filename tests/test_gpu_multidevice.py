"""`-m gpu`, boxes with MORE THAN ONE GPU only (skipped -- with the reason in the run's tail -- on the one-GPU lease the
builder gets): the first run on a multi-GPU node exercises, without any new work,
  * femto_amd_comm_gather with N > 1 ranks (the grouped ncclSend / ncclRecv gather of SURVEY.md 8(e), one rank per GPU);
  * a range-split index whose parts live on DIFFERENT devices, so that a "block fault" is a real peer load over xGMI
    (BASELINE configs[4]), in one process and across processes;
  * a replicated multi-device handle and a striped one over real devices.
Everything here has run with all "peers" on one device (tests/test_gpu_parity.py); nothing new is built for it."""
import os
import threading

import numpy as np
import pytest

import femto_amd

pytestmark = pytest.mark.gpu


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:      # noqa: BLE001
        return 0


need2 = pytest.mark.skipif(_ngpus() < 2, reason="needs >= 2 GPUs: %d visible (multi-GPU paths of SURVEY.md 8(e) NOT exercised on real peers)" % _ngpus())


@need2
def test_comm_gather_n_ranks(fixtures):
    """one rank per GPU (threads of this process, each with its own handle and communicator rank): every rank's buffer arrives
    at its slot on the root, for two roots and two payload sizes"""
    import torch
    n = min(_ngpus(), 8)
    fx = fixtures("acgt48k")
    uid = femto_amd.Index.comm_unique_id()
    errors, got = [], {}

    def rank(r):
        try:
            torch.cuda.set_device(r)
            ix = femto_amd.Index(fx.index, device=r)
            ix.comm_init(uid, n, r)
            assert ix.comm_info() == {"nranks": n, "rank": r}
            for root, count in ((0, 1000), (n - 1, 1 << 20)):
                src = torch.arange(count, dtype=torch.int64, device=f"cuda:{r}") * (r + 1) + r
                dst = torch.zeros(count * n if r == root else 1, dtype=torch.int64, device=f"cuda:{r}")
                st = torch.cuda.current_stream(r).cuda_stream
                ix.comm_gather(src.data_ptr(), dst.data_ptr() if r == root else 0, count * 8, root, st)
                torch.cuda.synchronize(r)
                if r == root:
                    got[(root, count)] = dst.cpu().numpy()
            ix.close()
        except Exception as ex:      # noqa: BLE001
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    for (root, count), arr in got.items():
        want = np.concatenate([np.arange(count, dtype=np.int64) * (r + 1) + r for r in range(n)])
        assert np.array_equal(arr, want), (root, count)
    assert len(got) == 2


@need2
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc", "runs3doc"])
def test_range_split_two_devices(fixtures, name):
    """part p of the index on GPU p: the lane kernels of either part read the other part's segment lines and block images
    as peer loads; leaf requests, count and locate equal the reference's goldens from BOTH parts"""
    fx = fixtures(name)
    g = fx.gold
    nparts = min(_ngpus(), 4)
    parts = [femto_amd.Index(fx.index, device=p, part=p, nparts=nparts) for p in range(nparts)]
    for a in parts:
        for b in parts:
            if a is not b:
                a.split_attach_local(b)
    for a in parts:
        a.split_commit()
    plen, flat, starts = fx.patterns
    rows = np.arange(parts[0].info.total_length, dtype=np.int64)
    for ix in parts:
        ch, occ, off = ix.block_requests(rows)
        assert np.array_equal(ch, g["L"]) and np.array_equal(occ, g["occ"]) and np.array_equal(off, g["off"])
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, g["count_first"]) and np.array_equal(last, g["count_last"])
        for mo, noccs, offs in fx.locate_cases():
            k, got = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(k, noccs) and np.array_equal(got, offs), mo
    for ix in parts:
        ix.close()


@need2
def test_range_split_across_processes_on_two_devices(fixtures, tmp_path):
    """tests/split_worker.py with one GPU per rank: hipIpc handles of ANOTHER device's memory"""
    from gpu_common import _torchrun
    fx = fixtures("acgt48k")
    script = os.path.join(os.path.dirname(__file__), "split_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _torchrun(2, [script, fx.index, os.path.join(os.path.dirname(__file__), "golden", "acgt48k.npz"), str(tmp_path)], env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(2):
        assert (tmp_path / f"ok{r}").exists()


@need2
@pytest.mark.parametrize("striped", [False, True])
def test_multi_device_handle_on_real_devices(fixtures, striped):
    """femto_amd_open_multi / _open_multi_striped over distinct GPUs: host-pointer batches shard over them (replicated), or
    every GPU reads the other's stripes over xGMI (striped); answers equal the goldens"""
    for name in ("acgt48k", "eng2doc"):
        fx = fixtures(name)
        ix = femto_amd.Index(fx.index, devices=list(range(min(_ngpus(), 4))), striped=striped)
        plen, flat, starts = fx.patterns
        first, last = ix.count_flat(plen, flat, starts)
        assert np.array_equal(first, fx.gold["count_first"]) and np.array_equal(last, fx.gold["count_last"])
        for mo, noccs, offs in fx.locate_cases():
            k, got = ix.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(k, noccs) and np.array_equal(got, offs), (name, mo)
        ix.close()

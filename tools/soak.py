"""Randomised parity soak at medium sizes (run on the GPU box): python tools/soak.py [n_indexes] [seed0]
Random alphabets / run structure / document splits / index parameters at 0.3-6 M rows, every kernel family against
the oracle (count, locate with random clamps, leaf requests).  The committed test-suite runs the same comparison at
2-60 k rows (tests/test_gpu_parity.py::test_random_indexes_vs_oracle) and at fixed medium/full sizes."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import femto_amd  # noqa: E402
from femto_amd import textgen as tg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import gpu_common as gc  # noqa: E402  (device_locate, assert_row_free_equals)


def one(seed, root):
    rng = np.random.Generator(np.random.PCG64(777000 + seed))
    lo_n, hi_n = [int(x) for x in os.environ.get("FEMTO_AMD_SOAK_ROWS", "300000,6000000").split(",")]
    n = int(rng.integers(lo_n, hi_n))
    sigma = int(rng.choice([2, 4, 5, 7, 8, 9, 17, 33, 100, 200, 255, 256]))
    alphabet = rng.choice(256, sigma, replace=False).astype(np.uint8)
    if rng.random() < 0.4:
        runs = rng.geometric(1.0 / float(rng.choice([2, 20, 400])), n // 2 + 10)
        text = np.repeat(alphabet[rng.integers(0, sigma, len(runs))], runs)[:n]
    elif rng.random() < 0.5:       # skewed frequencies
        p = rng.dirichlet(np.full(sigma, 0.3))
        text = alphabet[rng.choice(sigma, n, p=p)]
    else:
        text = alphabet[rng.integers(0, sigma, n)]
    ndocs = int(rng.integers(1, 4))
    cuts = sorted(rng.choice(np.arange(1, len(text)), ndocs - 1, replace=False)) if ndocs > 1 else []
    docs = np.split(text, cuts)
    b_size = int(rng.choice([4096, 65536, 1 << 20]))
    block = b_size * int(rng.choice([1, 4, 16]))
    mark = int(rng.choice([1, 3, 5, 8, 20, 37]))
    params = f"block_size={block},bucket_size={b_size},chunk_size={b_size},mark_period={mark}"
    path = os.path.join(root, f"soak{seed}")
    femto_amd.build_index(path, docs, params=params, infos=[f"d{i}" for i in range(len(docs))], device=0)
    o = po.Oracle(path)
    os.environ["FEMTO_AMD_PACK2"] = "1"
    ix = femto_amd.Index(path, device=0)
    pats = []
    for _ in range(20000):
        ln = int(rng.integers(0, 40)) if rng.random() < 0.6 else int(rng.integers(40, 200))
        if rng.random() < 0.7 and len(text) > ln:
            s0 = int(rng.integers(0, len(text) - ln + 1))
            pats.append(tg.to_alpha(text[s0:s0 + ln]))
        else:
            pats.append(tg.to_alpha(alphabet[rng.integers(0, sigma, ln)]))
    plen, flat, starts = femto_amd.flatten(pats)
    of, ol = o.count_flat(plen, flat, starts, threads=16)
    mo = int(rng.integers(1, 30))
    on, oo = o.locate_flat(plen, flat, starts, mo, threads=16)
    # the index itself against the TEXT (a wrong suffix order would be self-consistent between GPU and oracle):
    # every located offset is an occurrence, sampled substrings are found, offsets of a pattern are distinct
    prepared = np.concatenate([np.concatenate([d.astype(np.uint16) + 5, [2]]) for d in docs])
    pos = np.concatenate([[0], np.cumsum(on)])
    for i in range(0, len(pats), 7):
        p_ = pats[i]
        got = oo[pos[i]:pos[i + 1]]
        assert len(set(got.tolist())) == len(got), (seed, "duplicate offsets")
        for off in got:
            assert np.array_equal(prepared[off:off + len(p_)], p_), (seed, i, "located offset is not an occurrence")
    rows = rng.integers(0, ix.info.total_length, 2000).astype(np.int64)
    want = [o.block_request(int(r), 7) for r in rows]
    info = ix.pack_info()
    modes = [m for m in (3, 4, 1, 0) if not (m == 3 and not info["available"]) and not (m == 4 and not info["available2"])]
    for mode in modes:
        ix.set_rank_mode(mode)
        f, l_ = ix.count_flat(plen, flat, starts)
        assert np.array_equal(f, of) and np.array_equal(l_, ol), (seed, mode, params, "count")
        nn, offs = ix.locate_flat(plen, flat, starts, mo)
        assert np.array_equal(nn, on) and np.array_equal(offs, oo), (seed, mode, params, "locate")
        ch, occ, off = ix.block_requests(rows)
        assert [(int(a), int(b), int(c)) for a, b, c in zip(ch, occ, off)] == want, (seed, mode, "leaf")
        # the one-call device chain, with rows and row-free (noccs + offsets as parallel_locate returns them)
        df, dl, dn, dst, do, dtot = gc.device_locate(ix, plen, flat, starts, mo, len(oo) + 16)
        assert dtot == len(oo) and np.array_equal(df, of) and np.array_equal(dl, ol) and np.array_equal(dn, on) and np.array_equal(do, oo), (seed, mode, "device chain")
        gc.assert_row_free_equals(ix, plen, flat, starts, mo, on, oo, (seed, mode, "row-free"))
    ix.close()
    for kw in (dict(dense_arrays=0), dict(hbm_budget_bytes=int(3 * n)), dict(hbm_budget_bytes=int(8 * n)), dict(tail_min=2, tail_ones=0)):      # sampled arrays, budgets, eager tails
        bx = femto_amd.Index(path, device=0, options=kw)
        if bx.rank_mode in (3, 4):
            f, l_ = bx.count_flat(plen, flat, starts)
            assert np.array_equal(f, of) and np.array_equal(l_, ol), (seed, kw, "count")
            gc.assert_row_free_equals(bx, plen, flat, starts, mo, on, oo, (seed, kw, "row-free"))
            df, dl, dn, dst, do, dtot = gc.device_locate(bx, plen, flat, starts, mo, len(oo) + 16)
            assert dtot == len(oo) and np.array_equal(df, of) and np.array_equal(dl, ol) and np.array_equal(do, oo), (seed, kw, "device chain")
        bx.close()
    if seed % 4 == 0:     # range-split three ways in this process: every part answers like the whole index
        parts = [femto_amd.Index(path, device=0, part=p_, nparts=3) for p_ in range(3)]
        for a in parts:
            for b in parts:
                if a is not b:
                    a.split_attach_local(b)
        for a in parts:
            a.split_commit()
        for a in parts:
            f, l_ = a.count_flat(plen, flat, starts)
            assert np.array_equal(f, of) and np.array_equal(l_, ol), (seed, "split count")
            nn, offs = a.locate_flat(plen, flat, starts, mo)
            assert np.array_equal(nn, on) and np.array_equal(offs, oo), (seed, "split locate")
        for a in parts:
            a.close()
    if seed % 4 == 1:     # host-pointer pipeline: a batch above the pipeline threshold, in both calling conventions,
        big = 300_000     # and one whose symbols overflow a pinned chunk (falls back to the plain path)
        sel = rng.integers(0, len(pats), big)
        bl = plen[sel]
        bs = starts[sel]                       # not monotone: patterns referenced out of order
        ix = femto_amd.Index(path, device=0)
        f, l_ = ix.count_flat(bl, flat, bs)
        assert np.array_equal(f, of[sel]) and np.array_equal(l_, ol[sel]), (seed, "pipeline count")
        nn, offs = ix.locate_flat(bl, flat, bs, mo)
        assert np.array_equal(nn, on[sel]), (seed, "pipeline locate")
        longp = tg.to_alpha(np.tile(text[:400], 1))
        lp = np.full(big, len(longp), dtype=np.int32)
        lflat = np.tile(longp, 2).astype(np.uint16)
        lst = rng.integers(0, len(longp), big).astype(np.int64)      # overlapping windows of one long string
        f2, l2 = ix.count_flat(lp, lflat, lst)
        f3, l3 = o.count_flat(lp, lflat, lst, threads=16)
        assert np.array_equal(f2, f3) and np.array_equal(l2, l3), (seed, "pipeline fallback count")
        ix.close()
    distinct = len(np.unique(text)) + 1
    print(f"seed {seed}: rows {n} sigma {distinct} docs {ndocs} {params} modes {modes} located {int(on.sum())} ok", flush=True)


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with tempfile.TemporaryDirectory() as td:
        for s in range(seed0, seed0 + count):
            one(s, td)
    print("soak ok")

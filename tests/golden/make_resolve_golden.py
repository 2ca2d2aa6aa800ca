#!/usr/bin/env python3
"""Golden vectors for offset -> (document, offset in document), captured THROUGH THE REFERENCE
(container only; needs /root/reference and `make -C oracle`):

    python tests/golden/make_resolve_golden.py

For every committed fixture index (tests/golden/<name>.tar.gz, built by the reference) `oracle/_ref/ref_tool resolve` calls
header_loc_request(HDR_LOC_RESOLVE_LOCATION) -- resolve_location, src/main/index.c:1587 -- for EVERY logical offset of the
index and HDR_LOC_REQUEST_DOC_LEN for every document; the answers go to tests/golden/resolve_golden.npz
(<name>_doc int32[n], <name>_off int64[n], <name>_len int64[ndocs]).  Only data is committed."""
import os
import sys
import tarfile
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import INDEX_FIXTURES  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


MANY_PARAMS = "block_size=65536,bucket_size=8192,mark_period=20"


def many_documents(seed=77, ndocs=5000):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(97, 101, int(rng.integers(0, 41))).astype(np.uint8) for _ in range(ndocs)]


def main():
    assert po.have_ref(), "build the reference first: make -C oracle"
    out = {}
    for name in INDEX_FIXTURES:
        with tempfile.TemporaryDirectory() as td:
            with tarfile.open(os.path.join(OUT, name + ".tar.gz")) as tf:
                tf.extractall(td)
            f = os.path.join(td, "resolve.bin")
            po.ref_tool("resolve", os.path.join(td, "index"), f)
            raw = np.fromfile(f, dtype=np.int64)
            n, ndocs = int(raw[0]), int(raw[1])
            pairs = raw[2:2 + 2 * n].reshape(n, 2)
            out[name + "_doc"] = pairs[:, 0].astype(np.int32)
            out[name + "_off"] = pairs[:, 1].copy()
            out[name + "_len"] = raw[2 + 2 * n:2 + 2 * n + ndocs].copy()
            assert len(out[name + "_len"]) == ndocs
            print(name, n, ndocs, int(pairs[:, 0].max()))
    # many documents (the fixtures hold at most three): 5000 documents of 0..40 bytes, built by the reference itself; the
    # test rebuilds the same index from the same seeded documents with the product's byte-identical builder
    with tempfile.TemporaryDirectory() as td:
        docs = many_documents()
        paths = []
        for i, d in enumerate(docs):
            paths.append(os.path.join(td, "d%05d" % i))
            d.tofile(paths[-1])
        po.ref_build(os.path.join(td, "index"), MANY_PARAMS, paths)
        f = os.path.join(td, "resolve.bin")
        po.ref_tool("resolve", os.path.join(td, "index"), f)
        raw = np.fromfile(f, dtype=np.int64)
        n, ndocs = int(raw[0]), int(raw[1])
        pairs = raw[2:2 + 2 * n].reshape(n, 2)
        out["manydocs_doc"] = pairs[:, 0].astype(np.int32)
        out["manydocs_off"] = pairs[:, 1].astype(np.int32)
        out["manydocs_len"] = raw[2 + 2 * n:2 + 2 * n + ndocs].copy()
        assert ndocs == len(docs) and n == sum(len(d) + 1 for d in docs)
        print("manydocs", n, ndocs)
    np.savez_compressed(os.path.join(OUT, "resolve_golden.npz"), **out)


if __name__ == "__main__":
    main()

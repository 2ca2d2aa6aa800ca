#!/usr/bin/env python3
"""Generate the committed golden fixtures from the GENUINE reference (container only).

Run from the repo root in the build container (needs /root/reference and `make -C oracle`):

    python tests/golden/make_golden.py

For every fixture it
  1. writes seeded synthetic documents (femto_amd/textgen.py),
  2. builds a femto index with the reference's own constructor via oracle/_ref/ref_tool
     (`build` -> index_documents, /root/reference/src/main/construct.c:572),
  3. captures golden vectors THROUGH THE REFERENCE API (`dump` -> block_request CHAR|OCCS|LOCATION
     per row, header C / block_occs; `occs`; `count` -> parallel_count; `locate` -> parallel_locate),
  4. stores `<name>.tar.gz` (documents + index files, i.e. data) and `<name>.npz` (vectors).

Only data is committed: index files, documents, input patterns and expected outputs.
"""
import os
import sys
import tarfile
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from femto_amd import textgen as tg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def patterns_for(docs, seed, n_hit=150, n_rand=100, kmax=24, alphabet=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    text = np.concatenate(docs)
    alphabet = alphabet if alphabet is not None else np.unique(text)
    pats = []
    for _ in range(n_hit):                       # P_hit: substrings of the text
        l = int(rng.integers(1, kmax + 1))
        l = min(l, len(text))
        s = int(rng.integers(0, len(text) - l + 1))
        pats.append(tg.to_alpha(text[s:s + l]))
    for _ in range(n_rand):                      # P_rand over the text's alphabet
        l = int(rng.integers(1, kmax + 1))
        pats.append(tg.to_alpha(rng.choice(alphabet, l)))
    # edge cases: empty pattern, absent character, whole first document,
    # pattern crossing SEOF, single characters incl. SEOF itself and code 260
    pats.append(np.zeros(0, dtype=np.uint16))
    pats.append(np.array([5 + 0], dtype=np.uint16))          # byte 0x00: absent
    pats.append(np.array([260], dtype=np.uint16))            # byte 0xff: absent, ch+1 == ALPHA_SIZE
    pats.append(np.array([2], dtype=np.uint16))              # SEOF alone
    pats.append(tg.to_alpha(docs[0][:2000]))                 # (prefix of) whole first document
    if len(docs) > 1:
        a = tg.to_alpha(docs[0][-3:])
        b = tg.to_alpha(docs[1][:3])
        pats.append(np.concatenate([a, np.array([2], dtype=np.uint16), b]))   # crosses SEOF
        pats.append(np.concatenate([a, np.array([2], dtype=np.uint16)]))      # ends at SEOF
    for c in alphabet[:6]:
        pats.append(tg.to_alpha(np.array([c], dtype=np.uint8)))
    return pats


def make_index_fixture(name, docs, params, seed, flatten=False, occ_chars=(), max_occs=(1, 3, 7, 1000), with_map=False):
    print("fixture", name)
    with tempfile.TemporaryDirectory() as td:
        docnames = []
        for i, d in enumerate(docs):
            fn = f"doc{i}.txt"
            np.asarray(d, dtype=np.uint8).tofile(os.path.join(td, fn))
            docnames.append(fn)
        cwd = os.getcwd()
        os.chdir(td)
        try:
            if with_map:   # document chunks in every bucket (the production femto_index layout)
                os.environ["FEMTO_REF_WITH_MAP"] = "1"
            po.ref_build("index", params, docnames)
            os.environ.pop("FEMTO_REF_WITH_MAP", None)
            if flatten:
                po.ref_tool("flatten", "index", "index.flat")
        finally:
            os.chdir(cwd)
        ipath = os.path.join(td, "index")
        d = po.ref_dump(ipath, os.path.join(td, "dump.bin"))
        gold = dict(C=d["C"], block_occs=d["block_occs"], L=d["L"], occ=d["occ"], off=d["off"])
        # LF^-1 per row through the reference's forward leaf requests (do_forward_query, server.c:2424)
        gold["fwd_ch"], gold["fwd_row"], gold["fwd_off"] = po.ref_forward(ipath, os.path.join(td, "fwd.bin"))
        for ch in occ_chars:
            po.ref_tool("occs", ipath, ch, os.path.join(td, "occs.bin"))
            gold[f"occs_ch{ch}"] = np.fromfile(os.path.join(td, "occs.bin"), dtype=np.int32)
        pats = patterns_for([np.asarray(x, dtype=np.uint8) for x in docs], seed)
        plen, flat, starts = po._flat(pats)
        gold["pat_len"], gold["pat_flat"] = plen, flat
        gold["count_first"], gold["count_last"] = po.ref_count(ipath, pats, td)
        for mo in max_occs:
            n, o = po.ref_locate(ipath, pats, mo, td)
            gold[f"loc{mo}_noccs"], gold[f"loc{mo}_offs"] = n, o
        # a clamp case that hits the reference's `last-first > max_occs` quirk exactly:
        cnt = gold["count_last"] - gold["count_first"] + 1
        quirk = [int(c) - 1 for c in cnt if 2 <= c <= 200][:3]
        for mo in quirk:
            if f"loc{mo}_noccs" not in gold:
                n, o = po.ref_locate(ipath, pats, mo, td)
                gold[f"loc{mo}_noccs"], gold[f"loc{mo}_offs"] = n, o
        gold["params"] = np.array(params)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **gold)
        with tarfile.open(os.path.join(OUT, name + ".tar.gz"), "w:gz") as tf:
            for fn in sorted(os.listdir(td)):
                if fn.startswith("doc") or fn in ("index", "index.flat"):
                    tf.add(os.path.join(td, fn), arcname=fn)


def make_bseq_kat():
    """The 12 sequences x 3 segment-type modes of /root/reference/src/main/wtree_test.c:440-580
    (large ones shortened to 8 KiB), encoded by the reference's bseq_construct_forcetype; the
    expected rank answers are the direct bit counts of the raw data (the test's own method)."""
    print("bseq KAT")
    rng = np.random.Generator(np.random.PCG64(4242))
    big = 8 * 1024
    str1 = b"abracadabradabrabadrafunzobomsemesaoedasamba"
    seqs = [
        (b"\x61\x7e\x33\x33\x33\x33\x33\x33\x33\x33", 80),
        (b"\x10\x20\x30\x40", 32),
        (b"\x11\x11\x21\x13\x31\x14\x64\x11\x5a\xa5\x10", 84),
        (str1, 8 * len(str1)),
        (rng.integers(0, 256, big, dtype=np.uint8).tobytes(), 8 * big),
        ((rng.integers(0, 26, big, dtype=np.uint8) + ord("a")).astype(np.uint8).tobytes(), 8 * big),
        (b"\x00" * big, 8 * big),
        (b"\xff" * big, 8 * big),
        (b"\x55" * big, 8 * big),
        (b"\x11" * big, 8 * big - 3),
        # long runs of varying length (RLE with large gamma codes) then noise then runs
        (np.packbits(np.concatenate([np.repeat(np.arange(40) % 2, rng.integers(1, 3000, 40)),
                                     rng.integers(0, 2, 3000),
                                     np.repeat(np.arange(9) % 2, rng.integers(500, 9000, 9))]).astype(np.uint8)).tobytes(), None),
        # sparse ones (geometric gaps)
        (np.packbits((rng.random(60000) < 0.01).astype(np.uint8)).tobytes(), 60000),
    ]
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, (raw, nbits) in enumerate(seqs):
            if nbits is None:
                nbits = 8 * len(raw) - 5
            rp = os.path.join(td, "raw.bin")
            open(rp, "wb").write(raw)
            out[f"s{i}_raw"] = np.frombuffer(raw, dtype=np.uint8)
            out[f"s{i}_nbits"] = np.array(nbits)
            for typ in (-1, 0, 1):
                zp = os.path.join(td, "z.bin")
                po.ref_tool("bseq", rp, nbits, typ, zp)
                out[f"s{i}_t{typ + 1}_z"] = np.fromfile(zp, dtype=np.uint8)
    out["nseq"] = np.array(len(seqs))
    np.savez_compressed(os.path.join(OUT, "bseq_kat.npz"), **out)


def main():
    if not po.have_ref():
        po.build()
    assert po.have_ref(), "reference binary missing (needs /root/reference)"
    A = ord("A") + 5
    make_index_fixture("acgt48k", [tg.t_acgt(49152, 11)], "block_size=16384,bucket_size=4096,mark_period=20",
                       seed=101, flatten=True, occ_chars=(A, ord("T") + 5, 2, 5 + ord("N")))
    eng = tg.t_eng(40000, 12)
    make_index_fixture("eng2doc", [eng[:25000], eng[25000:]], "block_size=32768,bucket_size=8192,mark_period=16",
                       seed=102, occ_chars=(5 + ord(" "), 5 + ord("e"), 5 + ord("\n")))
    make_index_fixture("counter400_small", [tg.t_counter(400)], "block_size=16,bucket_size=4,chunk_size=2,mark_period=20",
                       seed=103, occ_chars=(5 + ord("a"),))
    make_index_fixture("counter400_default", [tg.t_counter(400)], "-", seed=104, occ_chars=(5 + ord("f"),))
    rng = np.random.Generator(np.random.PCG64(77))
    runs = np.concatenate([np.full(6000, ord("a")), np.full(3000, ord("b")), np.tile([ord("a"), ord("b")], 2000),
                           np.full(1, ord("c")), np.full(3000, ord("a")),
                           rng.choice(np.frombuffer(b"ab", dtype=np.uint8), 3000, p=[0.97, 0.03])]).astype(np.uint8)
    make_index_fixture("runs3doc", [runs[:9000], np.array([ord("z")], dtype=np.uint8), runs[9000:]],
                       "block_size=8192,bucket_size=2048,mark_period=7", seed=105,
                       occ_chars=(5 + ord("a"), 5 + ord("b")))
    make_index_fixture("construct_kat", [np.frombuffer(b"test_one;", dtype=np.uint8),
                                         np.frombuffer(b"test_two_fun;", dtype=np.uint8)],
                       "mark_period=100", seed=106, occ_chars=(5 + ord("t"), 5 + ord("n"), 5 + ord("e")))
    rng2 = np.random.Generator(np.random.PCG64(78))
    cd = [rng2.choice(np.frombuffer(b"abcdefgh \n", dtype=np.uint8), k).astype(np.uint8) for k in (3000, 2500)]
    make_index_fixture("chunks2doc", cd, "block_size=2048,bucket_size=2048,chunk_size=256,mark_period=10", seed=107,
                       occ_chars=(5 + ord("a"),), with_map=True)
    # the reference tests' "big buckets" parameter set (index_test_funcs.c:55-65): bucket size NOT a power of two
    rng3 = np.random.Generator(np.random.PCG64(79))
    b1000 = np.concatenate([tg.t_counter(3000), rng3.choice(np.frombuffer(b"abcdef", dtype=np.uint8), 4500)]).astype(np.uint8)
    make_index_fixture("b1000", [b1000], "block_size=10000,bucket_size=1000,chunk_size=1000,mark_period=20", seed=108,
                       occ_chars=(5 + ord("a"), 5 + ord("f")))
    # all 256 byte values (deep Huffman codes, every inUse16 group set) in two documents
    by = rng3.integers(0, 256, 12000, dtype=np.uint8)
    make_index_fixture("bytes256", [by[:7000], by[7000:]], "block_size=8192,bucket_size=4096,mark_period=9", seed=109,
                       occ_chars=(5, 260, 5 + 128))
    make_bseq_kat()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden dir bytes:", tot)


if __name__ == "__main__":
    main()

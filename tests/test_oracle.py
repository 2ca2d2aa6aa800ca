"""CPU (`-m "not gpu"`) tests that PIN THE ORACLE (oracle/femto_oracle.c):
  * the reference's own known-answer tests restated as data
    (/root/reference/src/main/wtree_test.c:286-335, :440-580; index_test.c:507-734),
  * golden vectors captured from the genuine reference on committed fixture indexes,
  * brute force over the fixture texts (the method of index_test.c:351-434).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, INDEX_FIXTURES
from oracle import pyoracle as po


def test_gamma_known_answers():
    # wtree_test.c:320-334: gamma(1..5) left-aligned in a 64-bit word
    kat = {1: 0x8000000000000000, 2: 0x4000000000000000, 3: 0x6000000000000000,
           4: 0x2000000000000000, 5: 0x2800000000000000}
    for v, word in kat.items():
        out = C.c_uint()
        nbits = po.lib().fo_decode_gamma(word, C.byref(out))
        assert out.value == v
        assert nbits == 2 * (v.bit_length() - 1) + 1
    # round trip over a spread of values (wtree_test.c:302-318)
    for v in [1, 2, 3, 7, 8, 100, 511, 512, 65535, 1 << 20, (1 << 31) - 1]:
        k = v.bit_length() - 1
        word = v << (64 - (2 * k + 1))
        out = C.c_uint()
        assert po.lib().fo_decode_gamma(word, C.byref(out)) == 2 * k + 1
        assert out.value == v


def test_varbyte_round_trip():
    # encode_varbyte semantics (wtree_funcs.h:437-454): LE 7-bit groups, last byte has 0x80
    for v in [0, 1, 127, 128, 255, 300, 16383, 16384, 1 << 20, (1 << 32) - 1]:
        enc = bytearray()
        x = v
        while True:
            w = x & 0x7F
            x >>= 7
            if x == 0:
                enc.append(w | 0x80)
                break
            enc.append(w)
        out = C.c_uint()
        n = po.lib().fo_decode_varbyte(bytes(enc) + b"\0" * 8, C.byref(out))
        assert n == len(enc) and out.value == v


def test_bseq_rank_exhaustive_on_reference_encoded_sequences():
    """wtree_test.c:440-580: every index of 12 sequences x 3 segment-type modes."""
    kat = np.load(os.path.join(GOLDEN, "bseq_kat.npz"))
    for i in range(int(kat["nseq"])):
        raw = kat[f"s{i}_raw"]
        nbits = int(kat[f"s{i}_nbits"])
        bits = np.unpackbits(raw)[:nbits].astype(np.int64)
        ones = np.cumsum(bits)
        zeros = np.arange(1, nbits + 1) - ones
        for t in range(3):
            got = po.bseq_rank_all(kat[f"s{i}_t{t}_z"], nbits)
            assert np.array_equal(got[:, 0], zeros), (i, t)
            assert np.array_equal(got[:, 1], ones), (i, t)
            assert np.array_equal(got[:, 2], bits), (i, t)


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_rows_match_reference(fixtures, name):
    fx = fixtures(name)
    g = fx.gold
    o = po.Oracle(fx.index)
    n = o.total_length
    assert n == len(g["L"]) == sum(len(d) + 1 for d in fx.docs)
    for ch in range(262):
        assert o.C(ch) == g["C"][ch]
    for ch in range(261):
        for b in range(o.num_blocks):
            assert o.block_occs(ch, b) == g["block_occs"][ch, b]
    step = 1 if n <= 20000 else 7
    for row in list(range(0, n, step)) + [n - 1]:
        ch, occ, off = o.block_request(row, 7)
        assert (ch, occ, off) == (g["L"][row], g["occ"][row], g["off"][row]), row
    for key in g.files:
        if key.startswith("occs_ch"):
            ch = int(key[7:])
            want = g[key]
            for row in list(range(0, n, step)) + [n - 1]:
                assert o.block_request(row, 2, ch)[1] == want[row], (ch, row)
    o.close()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_count_and_locate_match_reference(fixtures, name):
    fx = fixtures(name)
    o = po.Oracle(fx.index)
    plen, flat, starts = fx.patterns
    first, last = o.count_flat(plen, flat, starts)
    assert np.array_equal(first, fx.gold["count_first"])
    assert np.array_equal(last, fx.gold["count_last"])
    f2, l2 = o.count_flat(plen, flat, starts, threads=3)
    assert np.array_equal(first, f2) and np.array_equal(last, l2)
    for mo, noccs, offs in fx.locate_cases():
        n, got = o.locate_flat(plen, flat, starts, mo, threads=2)
        assert np.array_equal(n, noccs), mo
        assert np.array_equal(got, offs), mo
    o.close()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_count_locate_brute_force(fixtures, name):
    """index_test.c:351-434: counts vs brute force, located offsets verified against the text."""
    fx = fixtures(name)
    text = fx.prepared_text()
    o = po.Oracle(fx.index)
    plen, flat, starts = fx.patterns
    first, last = o.count_flat(plen, flat, starts)
    noccs, offs = o.locate_flat(plen, flat, starts, 1 << 30)
    pos = 0
    for i in range(len(plen)):
        p = flat[starts[i]:starts[i] + plen[i]]
        cnt = max(0, last[i] - first[i] + 1)
        if len(p) == 0:
            assert cnt == len(text)
        elif (p[:-1] == 2).any():
            # a pattern that CONTINUES past a SEOF: the reference's LF mapping of SEOF rows follows
            # the circular convention L[suffix 0] = SEOF (bwt_qsufsort.c:62-83), so which document
            # start it lands on is "indeterminate" (server.c:2336-2342); parity with the reference is
            # what the golden tests check, brute force over the concatenation does not apply.
            pass
        else:
            w = np.lib.stride_tricks.sliding_window_view(text, len(p)) if len(text) >= len(p) else np.zeros((0, len(p)))
            hits = np.nonzero((w == p).all(axis=1))[0]
            assert cnt == len(hits), i
            assert sorted(offs[pos:pos + noccs[i]]) == list(hits), i
        pos += noccs[i]
    o.close()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_forward_steps_match_reference(fixtures, name):
    """do_forward_query's leaf requests (LF^-1: bsearch_C, bsearch_block_occs, block_request_row with
    wtree_select / bseq_select, LOCATION) against the reference's answers for the same rows."""
    fx = fixtures(name)
    g = fx.gold
    o = po.Oracle(fx.index)
    n = o.total_length
    step = 1 if n <= 6000 else 11
    for row in list(range(0, n, step)) + [n - 1]:
        assert o.forward_step(row) == (g["fwd_ch"][row], g["fwd_row"][row], g["fwd_off"][row]), row
        ch, nr, _ = o.back_step(row)            # LF^-1(LF(row)) == row unless the walk stops at SEOF
        if nr >= 0:
            assert o.forward_step(nr)[1] == row
    o.close()


def test_flattened_index_equals_directory(fixtures):
    fx = fixtures("acgt48k")
    a, b = po.Oracle(fx.index), po.Oracle(fx.flat)
    plen, flat, starts = fx.patterns
    assert all(np.array_equal(x, y) for x, y in zip(a.count_flat(plen, flat, starts), b.count_flat(plen, flat, starts)))
    assert all(np.array_equal(x, y) for x, y in zip(a.locate_flat(plen, flat, starts, 5), b.locate_flat(plen, flat, starts, 5)))


def test_construct_known_answers(fixtures):
    """The hard-coded constants of test_construct (index_test.c:594-700): two documents
    "test_one;" and "test_two_fun;", mark period 100."""
    fx = fixtures("construct_kat")
    o = po.Oracle(fx.index)
    co = 5
    assert o.block_request(7, 1)[0] == co + ord("n")
    assert o.block_request(19, 2, co + ord("n"))[1] == 2
    assert o.block_request(1, 2, co + ord("e"))[1] == 0
    ch, occ, _ = o.block_request(8, 3)
    assert (ch, occ) == (co + ord("t"), 3)
    assert o.block_request(8, 2, co + ord("t"))[1] == 3
    # rows 19/20: 't' rows whose suffix starts a document -> marked with that document's start
    ch, occ, off = o.block_request(19, 7)
    assert o.block_request(19, 2, co + ord("t"))[1] == 4 and off == 0
    assert o.block_request(21, 7)[2] == -1
    assert o.block_request(20, 7)[2] == len(b"test_one;") + 1
    doc, doff = C.c_int64(), C.c_int64()
    po.lib().fo_resolve_location(o.h, len(b"test_one;") + 1, C.byref(doc), C.byref(doff))
    assert (doc.value, doff.value) == (1, 0)
    assert o.C(255) == o.total_length == len(b"test_one;") + len(b"test_two_fun;") + 2
    assert o.C(co + ord("e")) == 7
    assert o.C(co + ord("t")) == 17

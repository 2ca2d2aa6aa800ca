"""Regular-expression / automaton search (do_regexp_query src/main/server.c:1656).

Parity is pinned at the automaton: tests/golden/<fixture>_regexp.npz holds automata (the flat form of the reference's
nfa_description_t) and the sorted result lists the GENUINE do_regexp_query returned for them through
setup_regexp_query_take_nfa (tests/golden/make_regexp_golden.py, oracle/ref_tool.c `regexp_nfa`).  Checked here:
  * CPU: the parser + Thompson construction against Python's `re.fullmatch`; the position automaton built from every stored
    pattern text equals the stored automaton; the oracle's restatement of do_regexp_query equals the reference's lists;
  * GPU: femto_amd_nfa_search_batch (one workgroup per automaton, the whole fixture's automata in ONE call) equals the
    reference's lists bit for bit on every rank layout; fresh random automata against the genuine reference run on the GPU
    box; the iteration limit (ERR_OVERWORKED) and the stack bound against the restatement."""
import os
import re

import numpy as np
import pytest

import femto_amd
from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REGEXP_FIXTURES = ["acgt48k", "eng2doc", "bytes256", "runs3doc", "chunks2doc"]


def load_regexp_golden(name):
    """[(Nfa, regex bytes or None, approx tuple, (err_code, first, last, len, cost))] of one fixture"""
    g = np.load(os.path.join(GOLDEN, name + "_regexp.npz"))
    out, ts, tt, tn, rr = [], 0, 0, 0, 0
    for i in range(int(g["n"])):
        n = int(g["num_nodes"][i])
        tstart = g["trans_start"][ts:ts + n + 1]
        t = int(tstart[-1])
        a = femto_amd.Nfa(tstart, g["trans_char"][tt:tt + t], g["trans_dest"][tt:tt + t], g["is_start"][tn:tn + n],
                          g["is_final"][tn:tn + n], tuple(int(v) for v in g["settings"][i]))
        c = int(g["res_count"][i])
        want = (int(g["err_code"][i]), g["res_first"][rr:rr + c], g["res_last"][rr:rr + c], g["res_len"][rr:rr + c], g["res_cost"][rr:rr + c])
        out.append((a, bytes(g["regex"][i]) if g["from_regex"][i] else None, tuple(int(v) for v in g["approx"][i]), want))
        ts += n + 1
        tt += t
        tn += n
        rr += c
    return out


def _same(got, want):
    return got[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(got[1:], want[1:]))


def _rand_regex(rng, depth=0):
    """a random pattern over {a,b,c} in the common subset of femto's syntax and Python's"""
    r = rng.random()
    if depth > 3 or r < 0.35:
        k = rng.integers(0, 6)
        return [b"a", b"b", b"c", b".", b"[ab]", b"[^a]"][k]
    if r < 0.55:
        return _rand_regex(rng, depth + 1) + _rand_regex(rng, depth + 1)
    if r < 0.75:
        return b"(" + _rand_regex(rng, depth + 1) + b"|" + _rand_regex(rng, depth + 1) + b")"
    op = [b"*", b"+", b"?"][rng.integers(0, 3)]
    return b"(" + _rand_regex(rng, depth + 1) + b")" + op


def test_nfa_matches_like_a_regex_library():
    rng = np.random.Generator(np.random.PCG64(7))
    checked = 0
    for _ in range(300):
        rx = _rand_regex(rng)
        py = re.compile(b"(?s)" + rx)
        for _ in range(40):
            s = bytes(rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=int(rng.integers(0, 7))))
            got = femto_amd.regexp_match(rx, s)
            assert got is not None, rx
            assert got == (py.fullmatch(s) is not None), (rx, s)
            checked += 1
    assert checked == 12000


def _accepts_reversed(a, s):
    """does the epsilon-free automaton of the reversed pattern accept s read from its last byte to its first?"""
    cur = {i for i in range(a.num_nodes) if a.is_start[i]}
    for b in reversed(s):
        nxt = set()
        for i in cur:
            for e in range(a.trans_start[i], a.trans_start[i + 1]):
                if a.trans_char[e] == b + 5:
                    nxt.add(int(a.trans_dest[e]))
        cur = nxt
    return any(a.is_final[i] for i in cur)


def test_reversed_position_automaton_accepts_the_same_language():
    rng = np.random.Generator(np.random.PCG64(8))
    for _ in range(200):
        rx = _rand_regex(rng)
        py = re.compile(b"(?s)" + rx)
        a = femto_amd.Nfa.from_regex(rx)
        for _ in range(30):
            s = bytes(rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=int(rng.integers(0, 7))))
            assert _accepts_reversed(a, s) == (py.fullmatch(s) is not None), (rx, s)


def test_pattern_syntax_of_query_format_txt():
    """(the token rules and the grammar, one by one: tests/test_query_language.py)"""
    m = femto_amd.regexp_match
    assert m(rb"black sheep", b"blacksheep") is True          # unescaped whitespace separates terms (QUERY_FORMAT.txt)
    assert m(rb"black\ sheep", b"black sheep") is True
    assert m(rb'"a b"c', b"a bc") is True and m(rb"'a\n'", b"a\\n") is True
    assert m(rb"\x41\n", b"A\n") is True and m(rb"\.", b".") is True and m(rb"\.", b"x") is False
    assert m(rb"[a-c]+x?", b"abca") is True and m(rb"[\]a]", b"]") is True and m(rb"[^\n]", b"\n") is False
    assert m(rb"(ab)*", b"ababab") is True and m(rb"(ab)*", b"aba") is False
    # the reference's grammar has no empty sequence and one repeat operator per atom (posix.bison.y:88-105)
    for bad in (rb"(a", rb"a)", rb"[a", rb"*a", rb"a|", rb"[]a]", b'"a'):
        assert m(bad, b"a") is None, bad


def test_hostile_patterns_are_refused_not_crashed():
    """nesting deeper than the parser's bound, automata beyond 4096 states, megabytes of pattern text: an error code"""
    m = femto_amd.regexp_match
    assert m(b"(" * 200000 + b"a" + b")" * 200000, b"a") is None
    assert m(b"(" * 200 + b"a" + b")" * 200, b"a") is True
    assert m(b'"' + b"a" * 100000 + b'"', b"a") is None
    assert m(b"a" * 3000000, b"a") is None
    with pytest.raises(femto_amd.FemtoAmdError):
        femto_amd.Nfa.from_regex(b"(" * 100000 + b"a")
    # APPROX settings as compile_regexp_from_ast checks them (compile_regexp.c:673-685)
    for bad in [(3, 1, 1, 1), (3, 2, 1, 1), (-1, 1, 1, 1), (1, 0, 1, 1), (300, 100, 100, 100)]:
        with pytest.raises(femto_amd.FemtoAmdError):
            femto_amd.Nfa.from_regex(b"abc", bad)
    assert femto_amd.Nfa.from_regex(b"abc", (2, 1, 1, 1)).settings == (3, 1, 1, 1)
    assert femto_amd.Nfa.from_regex(b"abc", (1, 2, 1, 2)).settings == (2, 2, 1, 2)


def test_malformed_automata_are_refused(fixtures):
    """femto_amd_nfa_search_batch validates every automaton before anything reaches the device: destinations and characters in
    range, monotone transition starts, costs and cost_bound 1..255 (errors are counted in one byte), at most 2048 nodes"""
    fx = fixtures("acgt48k")
    ix = femto_amd.Index(fx.index, device=-1)            # parse-only handle: validation comes first, then "no device"
    good = femto_amd.Nfa([0, 1, 1], [70], [1], [1, 0], [0, 1])
    with pytest.raises(femto_amd.FemtoAmdError) as e:
        ix.nfa_search_batch([good])
    assert e.value.code == femto_amd.ERR_INVALID           # well-formed: only the missing device stops it
    bad = [femto_amd.Nfa([0, 1, 1], [70], [2], [1, 0], [0, 1]),            # destination out of range
           femto_amd.Nfa([0, 1, 1], [261], [1], [1, 0], [0, 1]),           # character >= ALPHA_SIZE
           femto_amd.Nfa([0, 1, 1], [-1], [1], [1, 0], [0, 1]),
           femto_amd.Nfa([0, 2, 1], [70], [1], [1, 0], [0, 1]),            # starts not monotone / not spanning
           femto_amd.Nfa([1, 1, 1], [70], [1], [1, 0], [0, 1]),
           femto_amd.Nfa([0, 1, 1], [70], [1], [1, 0], [0, 1], (0, 1, 1, 1)),      # cost_bound < 1
           femto_amd.Nfa([0, 1, 1], [70], [1], [1, 0], [0, 1], (256, 1, 1, 1)),
           femto_amd.Nfa([0, 1, 1], [70], [1], [1, 0], [0, 1], (2, 0, 1, 1)),      # a cost < 1
           femto_amd.Nfa([0, 1, 1], [70], [1], [1, 0], [0, 1], (2, 1, 1, 300)),
           femto_amd.Nfa([0] * 2050, [], [], [1] * 2049, [0] * 2049)]              # too many nodes
    for a in bad:
        with pytest.raises(femto_amd.FemtoAmdError) as e:
            ix.nfa_search_batch([good, a])
        assert e.value.code == femto_amd.ERR_PARAM, (a.trans_start[:3], a.settings)
    ix.close()


@pytest.mark.parametrize("name", REGEXP_FIXTURES)
def test_compiled_automata_equal_the_golden_ones(name):
    """the construction (parser -> Thompson -> position automaton of the reversed pattern) is deterministic and pinned:
    the reference's result lists in the goldens were produced for exactly these automata"""
    n = 0
    for a, rx, approx, _ in load_regexp_golden(name):
        if rx is None:
            continue
        # (the reference's grammar has no empty query: the pattern that matches the empty string is written '')
        b = femto_amd.Nfa.from_regex(rx or b"''", approx)
        assert b.settings == a.settings and b.num_nodes == a.num_nodes, rx
        for f in ("trans_start", "trans_char", "trans_dest", "is_start", "is_final"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (rx, f)
        n += 1
    assert n >= 3


@pytest.mark.parametrize("name", REGEXP_FIXTURES)
def test_oracle_restatement_equals_reference_goldens(fixtures, name):
    fx = fixtures(name)
    o = po.Oracle(fx.index)
    cases = load_regexp_golden(name)
    for i, (a, rx, approx, want) in enumerate(cases):
        assert _same(o.nfa_search(a), want), (name, i, rx, approx)
    o.close()
    assert len(cases) >= 70


def _check_batch(ix, cases, what):
    start, first, last, mlen, cost, status = ix.nfa_search_batch([c[0] for c in cases])
    assert len(status) == len(cases) and start[-1] == len(first)
    for i, (a, rx, approx, want) in enumerate(cases):
        s, e = int(start[i]), int(start[i + 1])
        got = (int(status[i]), first[s:e], last[s:e], mlen[s:e], cost[s:e])
        assert _same(got, want), (what, i, rx, approx, a.settings, got[0], want[0], e - s, len(want[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", REGEXP_FIXTURES)
def test_nfa_search_batch_equals_reference_goldens(fixtures, name):
    """every automaton of the fixture in ONE call, on every rank layout the handle has"""
    import torch
    assert torch.cuda.is_available()
    fx = fixtures(name)
    cases = load_regexp_golden(name)
    ix = femto_amd.Index(fx.index, device=0)
    _check_batch(ix, cases, (name, "default mode %d" % ix.rank_mode))
    if ix.rank_mode != 1:
        ix.set_rank_mode(1)                     # femto's own wavelet tree
        _check_batch(ix, cases, (name, "mode 1"))
    ix.close()
    if os.path.exists(fx.flat):                 # the flattened container is the same index
        ix = femto_amd.Index(fx.flat, device=0)
        _check_batch(ix, cases[:20], (name, "flat"))
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(FEMTO_AMD_NFA_LPT="0"), dict(FEMTO_AMD_NFA_FAIR="0"), dict(FEMTO_AMD_NFA_LPT="0", FEMTO_AMD_NFA_FAIR="0", FEMTO_AMD_NFA_STATS="1")])
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_nfa_batch_order_and_fair_share_do_not_change_results(fixtures, monkeypatch, name, env):
    """the batch handed out in the caller's order instead of longest-predicted-first, and without the fair share between concurrent
    kernels (the A/B knobs of profiles/r06_regexp_concurrent.txt): the same golden result lists"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fx = fixtures(name)
    cases = load_regexp_golden(name)
    ix = femto_amd.Index(fx.index, device=0)
    _check_batch(ix, cases, (name, env))
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_concurrent_callers_share_the_workgroups(fixtures, name):
    """Several threads call femto_amd_nfa_search_batch on ONE handle at once (the reference keeps many do_regexp_query state
    machines in flight, src/main/server.c:3969-4001): every caller's lists equal the reference's goldens -- the batch is handed
    out longest-predicted-first and a kernel gives up workgroups beyond its fair share between automata, and neither may change a
    result list; femto_amd_nfa_stats reports what the calling thread's batch did."""
    import threading
    fx = fixtures(name)
    cases = load_regexp_golden(name)
    ix = femto_amd.Index(fx.index, device=0)
    errors, stats = [], {}

    def caller(t):
        try:
            mine = cases[t::3] + cases[:5]
            for rep in range(3):
                _check_batch(ix, mine, (name, "thread", t, rep))
            stats[t] = ix.nfa_stats(thread=True)
        except Exception as ex:      # noqa: BLE001
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=caller, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join(600)
    assert not errors, errors
    for t in range(3):
        st = stats[t]
        assert st["automata"] == len(cases[t::3]) + 5 and st["pops"] >= st["pops_longest"] > 0 and st["workgroups"] >= 1, st
        assert 0 < st["busy_s"] and 0 < st["span_s"] < 60 and 0 < st["occupancy"] <= 1.0 + 1e-6, st
    ix.close()


def _random_acyclic(rng, alpha, approx):
    n = int(rng.integers(2, 12))
    ts, tc, td = [0], [], []
    for i in range(n):
        for _ in range(int(rng.integers(1, 6)) if i < n - 1 else 0):
            tc.append(int(rng.choice(alpha)))
            td.append(int(rng.integers(i + 1, n)))
        ts.append(len(tc))
    st = (rng.random(n) < 0.3).astype(np.uint8)
    st[0] = 1
    fi = (rng.random(n) < 0.2).astype(np.uint8)
    fi[n - 1] = 1
    se = (1, 1, 1, 1)
    if approx:
        b = int(rng.integers(2, 4))
        ins = max(int(rng.integers(1, 3)), (b + 2) // 3)
        se = (b, max(int(rng.integers(1, 3)), ins), int(rng.integers(1, 3)), ins)
    return femto_amd.Nfa(ts, tc, td, st, fi, se)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["acgt48k", "eng2doc"])
def test_fresh_automata_against_the_genuine_reference(fixtures, tmp_path, name):
    """automata nobody has seen before: the genuine do_regexp_query (oracle/_ref/ref_tool, on the GPU box's host) and the GPU
    must agree; a second, larger batch against the restatement"""
    import torch
    assert torch.cuda.is_available()
    fx = fixtures(name)
    rng = np.random.Generator(np.random.PCG64(int.from_bytes(os.urandom(4), "little")))
    chars = np.unique(np.concatenate(fx.docs)).astype(np.int32) + 5
    alpha = np.concatenate([chars[:8], chars[:8], np.array([2, 260], dtype=np.int32)])
    ix = femto_amd.Index(fx.index, device=0)
    if po.have_ref():
        nfas = [_random_acyclic(rng, alpha, k % 3 == 2) for k in range(120)]
        ref = po.ref_regexp_nfa(fx.index, nfas, str(tmp_path))
        _check_batch(ix, [(a, None, None, r) for a, r in zip(nfas, ref)], (name, "fresh vs reference"))
    o = po.Oracle(fx.index)
    nfas = [_random_acyclic(rng, alpha, k % 4 == 3) for k in range(1500)]
    _check_batch(ix, [(a, None, None, o.nfa_search(a)) for a in nfas], (name, "fresh vs restatement"))
    o.close()
    ix.close()


@pytest.mark.gpu
def test_iteration_limit_and_stack_bound(fixtures):
    """MAX_REGEXP_ITERATIONS (server.c:40,1821): beyond it the reference returns ERR_OVERWORKED and no results -- with the
    limit lowered, the GPU and the restatement must stop at the same pop; a stack bound the search outgrows is ERR_FULL"""
    import torch
    assert torch.cuda.is_available()
    fx = fixtures("eng2doc")
    ix = femto_amd.Index(fx.index, device=0)
    o = po.Oracle(fx.index)
    pats = [rb"(and|or)\ [a-z]+", rb"th[aeiou]+", rb"[a-z]+ing", rb"the"]
    nfas = [femto_amd.Nfa.from_regex(p) for p in pats]
    for limit in (3, 50, 400, 5000):
        ix.set_option("regexp_max_iterations", limit)
        start, first, last, mlen, cost, status = ix.nfa_search_batch(nfas)
        for i, a in enumerate(nfas):
            want = o.nfa_search(a, max_iterations=limit)
            s, e = int(start[i]), int(start[i + 1])
            assert _same((int(status[i]), first[s:e], last[s:e], mlen[s:e], cost[s:e]), want), (limit, pats[i])
        assert status[0] == femto_amd.ERR_OVERWORKED or limit >= 5000
    ix.set_option("regexp_max_iterations", 1000000)
    ix.set_option("regexp_stack_cap", 16)
    start, first, last, mlen, cost, status = ix.nfa_search_batch(nfas)
    assert status[0] == femto_amd.ERR_FULL and start[1] == start[0]
    assert status[3] == 0 and start[4] - start[3] == 1
    with pytest.raises(femto_amd.FemtoAmdError):
        ix.regexp_search(pats[0])
    ix.set_option("regexp_stack_cap", 1 << 18)
    # the search that outgrew the first arena is run again in a larger one: same results as unbounded
    first1, last1, mlen1 = ix.regexp_search(pats[0])
    want = o.nfa_search(nfas[0])
    assert np.array_equal(first1, want[1]) and np.array_equal(last1, want[2]) and np.array_equal(mlen1, want[3])
    # more results than the caller's arrays hold
    with pytest.raises(femto_amd.FemtoAmdError):
        ix.nfa_search_batch(nfas, max_results=2)
    # ... ERR_FULL leaves the exact number to call again with; max_results = 0 only counts (the raw ranges -- before the sort drops
    # ranges inside other results -- are buffered by the library, so neither depends on the caller's capacity)
    need = ix.last_total
    full = ix.nfa_search_batch(nfas, max_results=1 << 20)
    assert need == len(full[1]) > 2
    exact = ix.nfa_search_batch(nfas, max_results=need)
    assert all(np.array_equal(a, b) for a, b in zip(full, exact))
    cstart, cf, cl, cm, cc, cst = ix.nfa_search_batch(nfas, max_results=0)
    assert ix.last_total == need and len(cf) == 0 and np.array_equal(cstart, full[0]) and np.array_equal(cst, full[5])
    o.close()
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,patterns", [
    ("acgt48k", [rb"ACGTACG", rb"AC(GT|TG)+A", rb"G[AC]T[^A]GG", rb"TTT.TTT", rb"(ACG|TGCA)A?C", rb"GATTACA|TACAGAT"]),
    ("eng2doc", [rb"the", rb"th[aeiou]+", rb"e\.\ ", rb"q.", rb"\ (a|an|the)\ "]),
    ("bytes256", [rb"\x00.", rb"[\x80-\xff][\x00-\x10]", rb"\xfe|\xff\xff?"]),
])
def test_regexp_results_are_matches_in_the_text(fixtures, name, patterns):
    """what the results MEAN: every result is a string of the text that matches the pattern in full and has no proper
    suffix that matches (the search stops at the first final state), its range holds exactly the rows of the suffixes that
    start with it, and every minimal match of the text is covered by some result"""
    import torch
    assert torch.cuda.is_available()
    fx = fixtures(name)
    ix = femto_amd.Index(fx.index, device=0)
    text = b"\x00".join(d.tobytes() for d in fx.docs)     # same offsets as the prepared text (one separator per document)
    for rx in patterns:
        first, last, mlen = ix.regexp_search(rx)
        key = list(zip(first.tolist(), (-last).tolist()))
        assert key == sorted(key), rx                    # regexp_result_list_sort: first ascending, last descending
        assert (last >= first).all() and (mlen >= 1).all()
        py = re.compile(b"(?s)" + rx)
        covered = set()
        for f, l, m in zip(first, last, mlen):
            o = ix.locate_range(int(f), int(l))
            s = text[int(o[0]):int(o[0]) + int(m)]
            assert py.fullmatch(s) is not None, (rx, s)
            assert not any(py.fullmatch(s[k:]) for k in range(1, len(s))), (rx, s)
            assert all(text[int(p):int(p) + int(m)] == s for p in o[:50])
            covered.update(int(p) for p in o)
        # brute force: starts of minimal matches (a match none of whose proper suffixes matches), inside one document
        want, base = set(), 0
        for d in fx.docs:
            b = d.tobytes()
            for end in range(1, len(b) + 1):
                for st in range(end - 1, max(-1, end - 40), -1):
                    if py.fullmatch(b, st, end):
                        want.add(base + st)
                        break
            base += len(b) + 1
        assert want <= covered, (rx, len(want - covered))
    ix.close()

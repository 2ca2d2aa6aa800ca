"""femto's query language: the flex scanner and bison grammar of the reference (src/main/posix.flex.l,
src/main/posix.bison.y), streamline_query (src/main/query_planning.c) and simplify_query / icase_ast (src/main/ast.c) as
femto_amd/csrc/query_parser.hpp restates them.  flex/bison are not in the image, so the reference's parser cannot be run;
what pins the restatement:
  * the reference's OWN known answers for parse -> streamline -> print (src/main/query_planning_test.c:34-155), as data;
  * the whole query set of the reference's end-to-end test (src/test/test.pl, restated in tests/testpl.py): for every query
    the prepared automaton, simulated backwards over the documents the way do_regexp_query walks the index (stop at the
    first final state), must find exactly the offsets test.pl expects from Perl's regex engine (here: Python's);
  * the token rules one by one, each with the scanner line it restates.
CPU only -- the same queries run through femto_amd_search on the GPU in tests/test_gpu_cli.py."""
import re

import numpy as np
import pytest

import femto_amd
import testpl


# src/main/query_planning_test.c:34-155 -- (query, what ast_to_string prints after streamline_query)
STREAMLINE_KAT = [(b"a*(b|b?bb?)c", b"(b|bb?)c"), (b"a*(bc|d)+", b"(bc|d)"), (b"a*(bc|d)b?", b"(bc|d)"), (b"a?(b*cd|e?fg?)h", b"(cd|fg?)h"),
                  (b"a*b?cd(e*|f)g?", b"cd"), (b"a+bc+", b"abc"), (b"ab*c", b"ab*c")]


def test_streamline_known_answers_of_the_reference():
    for q, want in STREAMLINE_KAT:
        assert femto_amd.query_echo(q, streamline=True, usequotes=False) == want, q
    # ... and what those queries become: one string where nothing but a string is left (simplify_query)
    assert femto_amd.Nfa.from_query(b"a+bc+")[1].tolist() == [ord(c) + 5 for c in "abc"]
    assert femto_amd.Nfa.from_query(b"a*b?cd(e*|f)g?")[1].tolist() == [ord("c") + 5, ord("d") + 5]
    assert femto_amd.Nfa.from_query(b"ab*c")[1] is None
    # the first alternative of a TRAILING group keeps its optional tail (fix_final's loop stops at i > 0, query_planning.c:177)
    assert femto_amd.query_echo(b"x(ab?|cd?)") == b"x(ab?|c)"


def _accepts(q, s, **kw):
    return femto_amd.regexp_match(q, s)


def test_token_rules():
    m = femto_amd.regexp_match
    # whitespace separates terms (posix.flex.l:327), # starts a comment (:142), both vanish
    assert m(b"black sheep", b"blacksheep") and m(b"black # colour\nsheep", b"blacksheep") and not m(b"black sheep", b"black sheep")
    # \\c escapes (:113-127, :203-208), \\xNN (:196-201); one hex digit is not \\xNN: the characters x and 4
    assert m(rb"black\ sheep", b"black sheep") and m(rb"\x41\n\t\e", b"A\n\t\x1b") and m(rb"\.", b".") and not m(rb"\.", b"x")
    assert m(rb"\x4", b"x4") and m(rb"\x4g", b"x4g")
    # quotes (:144-186): '...' literal, "..." with escapes; a repeat applies to the WHOLE string token
    assert m(rb'"a b"c', b"a bc") and m(rb"'a\n'", b"a\\n") and m(rb'"a\"b\x41"', b'a"bA')
    assert m(rb"'ab'+", b"ababab") and not m(rb"'ab'+", b"abb")
    # {x hex} strings (:263-266; an odd digit is dropped, ast.c:56), {m} {m,} {m,n} (:267-270), any other { is a character
    assert m(b"{x 41 42}", b"AB") and m(b"{x4142}+", b"ABAB") and m(b"{x 41 4}", b"A") and m(b"{x}a", b"a")
    assert m(b"a{2}", b"aa") and not m(b"a{2}", b"aaa") and m(b"a{2,}", b"aaaaa") and not m(b"a{2,}", b"a")
    assert m(b"a{2,3}b", b"aaab") and not m(b"a{2,3}b", b"aaaab") and m(b"a{0}b", b"b") and m(b"a{3,2}", b"aaa") and not m(b"a{3,2}", b"aa")
    assert m(b"a{", b"a{") and m(b"a{,3}", b"a{,3}") and m(b"a}b]c-d,e", b"a}b]c-d,e") and m(b"{y}", b"{y}")
    # words (:285-318): three or more word characters are ONE string -- unless punctuation follows, which splits the last off
    assert m(b"abcd*", b"abc") and m(b"abcd*", b"abcddd") and not m(b"abcd*", b"abcdabcd")
    assert m(b"abcd *", b"abcdabcd") and m(b"abcd *", b"") and not m(b"abcd *", b"abcdd")
    assert m(b"ab*", b"abbb") and not m(b"ab*", b"abab") and m(b"abc+", b"abccc") and not m(b"abc+", b"abcabc")
    assert m("cafés?".encode(), "café".encode())          # bytes >= 0x80 are word characters
    # sets (:209-238): whitespace is literal inside, escapes work, ranges, negation within the 256 bytes (ast.c:324)
    assert m(b"[a-c]+x?", b"abca") and m(b"[ \\t]", b" ") and m(b"[ \\t]", b"\t") and m(rb"[\]\-\\]+", b"]-\\") and m(rb"[\x00-\x02]", b"\x01")
    assert not m(rb"[^\n]", b"\n") and m(rb"[^\n]", b"\x00") and m(b"[^a-y]", b"z") and not m(b"[^a-y]", b"b") and m(b"[z-a]?q", b"q")
    # . is any BYTE (period_range, ast.c:33)
    assert m(b"a.c", b"a\nc") and m(b"a.c", b"a\xffc")
    # one repeat operator per atom, sequences are never empty (posix.bison.y:88-105)
    for bad in (b"(a", b"a)", b"[a", b"*a", b"a**", b"a+?", b"a|", b"|a", b"()", b"[]", b"[]a]", b"[a-]", b"[-a]", b'"a', b"'a", b"", b"  ", b"a{5000}"):
        assert m(bad, b"a") is None, bad
    # the codes below the bytes (:188-194): \\x-03 is alpha 2, the end-of-document marker
    a, lit, _ = femto_amd.Nfa.from_query(rb"a\x-03")
    assert lit.tolist() == [ord("a") + 5, 2]


def test_keywords():
    q = femto_amd.Nfa.from_query
    # APPROX (:277-280) needs whitespace after its argument -- and, without an argument, TWO whitespace characters
    assert q(b"APPROX 1 black")[0].settings == (2, 1, 1, 1) and q(b"approx 2 black")[0].settings == (3, 1, 1, 1)
    assert q(b"APPROX 1:2:1:2 black")[0].settings == (2, 2, 1, 2) and q(b"APPROX  black")[0].settings == (2, 1, 1, 1)
    assert q(b"APPROX black")[1].tolist() == [c + 5 for c in b"APPROXblack"]
    assert q(b"APPROX 2x black")[1].tolist() == [c + 5 for c in b"APPROX2xblack"]
    assert q(b"APPROX 0 black")[1].tolist() == [c + 5 for c in b"black"]          # cost_bound 1: exact, and therefore one string
    assert q(b"APPROX 1 black")[1] is None and q(b"APPROX 1 black")[2] == b'"black"'
    for bad in (b"APPROX 3 black", b"APPROX 9:1:1:1 black"):                          # compile_regexp.c:673-685
        with pytest.raises(femto_amd.FemtoAmdError):
            q(bad)
    with pytest.raises(femto_amd.FemtoAmdError):
        q(b"black (APPROX 1 sheep)")                                                  # only at the start of a query
    # the boolean operators (:248-276) are recognised exactly where the scanner recognises them, and refused
    for bad in (b"black AND sheep", b"black or sheep", b"a NOT b", b"a THEN b", b"a then 20 b", b"a WITHIN 5 b"):
        with pytest.raises(femto_amd.FemtoAmdError) as e:
            q(bad)
        assert "boolean" in str(e.value)
    for fine, s in ((b"blackANDsheep", b"blackANDsheep"), (b"black AND", b"blackAND"), (b"sand or", b"sandor"), (b"a WITHIN b", b"aWITHINb"),
                    (b"'AND' x", b"ANDx"), (b"And x", b"Andx")):
        assert q(fine)[1].tolist() == [c + 5 for c in s], fine


def test_icase_and_echo():
    q = femto_amd.Nfa.from_query
    a, lit, echo = q(b"Ab1", icase=True)
    assert lit is None and echo == b"[Aa][Bb]1"
    assert q(b"[a-c]x", icase=True)[2] == b"[A-Ca-c][Xx]"
    # ast_to_string (ast.c:875-1120) with quotes: what femto_search --json echoes as "pattern"
    assert q(b"abc")[2] == b'"abc"' and q(b"ab")[2] == b'"ab"' and q(rb"a\*b\"")[2] == rb'"a*b\""' and q(b"abc(d|e)f")[2] == b'"ab"c(d|e)f'
    assert q(b"a.b[x-z]{2,3}c")[2] == b"a.b([x-z]){2,3}c" and q(rb"\x00\x01")[2] == rb'"\x00\x01"' and q(b"'a b'c|d")[2] == b'( "a b"c|d)'
    assert q(b"a+bc+")[2] == b'"abc"'                      # streamlined, simplified, then echoed


def _simulate(a, docs):
    """what do_regexp_query reports for the automaton `a` (of the reversed pattern), found by brute force: walking back from
    every end position of every document, the first string that reaches a final node is a result and is not extended"""
    start = [i for i in range(a.num_nodes) if a.is_start[i]]
    final = set(i for i in range(a.num_nodes) if a.is_final[i])
    trans = {}
    for i in range(a.num_nodes):
        for e in range(a.trans_start[i], a.trans_start[i + 1]):
            trans.setdefault((i, int(a.trans_char[e])), []).append(int(a.trans_dest[e]))
    out = []
    for di, d in enumerate(docs):
        offs = set()
        for end in range(1, len(d) + 1):
            cur = set(start)
            for k in range(end - 1, -1, -1):
                nxt = set()
                for s in cur:
                    nxt.update(trans.get((s, d[k] + 5), ()))
                cur = nxt
                if not cur:
                    break
                if cur & final:
                    offs.add(k)
                    break
        if offs:
            out.append((di, sorted(offs)))
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_the_reference_end_to_end_query_set(seed):
    docs, queries = testpl.make_docs_and_queries(seed)
    assert len(docs) == 36 and len(queries) > 400
    literal = regex = 0
    for fe, py in queries.items():
        a, lit, _ = femto_amd.Nfa.from_query(fe)
        want = testpl.expected_results(docs, py)
        if lit is not None:
            s = bytes((lit - 5).astype(np.uint8).tolist())
            got = [(i, [j for j in range(len(d) - len(s) + 1) if d[j:j + len(s)] == s]) for i, d in enumerate(docs)]
            got = [g for g in got if g[1]]
            literal += 1
        else:
            got = _simulate(a, docs)
            regex += 1
        assert got == want, (fe, py.pattern)
    assert literal > 300 and regex > 80


def test_behind_the_parser_equals_the_genuine_reference():
    """tests/golden/query_ast_golden.json (tests/golden/make_query_golden.py): 924 queries -- the reference's end-to-end test's
    whole query set, query_planning_test.c's strings, QUERY_FORMAT.txt's constructs, 400 random expressions -- whose trees, as
    THIS parser built them, were rebuilt with the reference's own constructors and run through the GENUINE streamline_query,
    simplify_query, icase_ast and ast_to_string (compiled from /root/reference by oracle/Makefile; only the flex/bison parser
    cannot be).  The product's restatements of those four must print the same, with and without --icase, with and without quotes."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "query_ast_golden.json")))
    assert len(g) >= 1800
    for e in g:
        q = e["query"].encode("latin-1")
        for usequotes, key in ((True, "quoted"), (False, "plain")):
            got = femto_amd.query_echo(q, streamline=True, simplify=True, icase=bool(e["icase"]), usequotes=usequotes)
            assert got is not None and got.decode("latin-1") == e[key], (q, e["icase"], key, got, e[key])

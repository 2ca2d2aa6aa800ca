"""`-m gpu`: femto_amd_search (tools/femto_amd_search.cpp), femto_search's counterpart, invoked the way
femto_search is invoked -- the pattern is a QUERY in femto's language, no extra flags -- and checked against brute force over
the documents, and against the reference's own end-to-end test (src/test/test.pl, restated in tests/testpl.py)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import femto_amd
import testpl
from femto_amd import build as b

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ok():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    b.build_tools()
    return True


def _run(args, **kw):
    return subprocess.run([b.SEARCH] + list(args), capture_output=True, timeout=300, **kw)


def _alphatos(s):
    """fprint_alpha (src/main/index_types.h:104-131): how femto_search prints a matched string"""
    out = b""
    for c in s:
        if c in (0x5c, 0x22):
            out += b"\\" + bytes([c])
        elif 32 <= c < 127:
            out += bytes([c])
        else:
            out += b"\\x%02x" % c
    return out


def _occurrences(docs, regex):
    """{matched string: [(doc, offset)]} of the strings do_regexp_query reports: walking back from every end position, the
    first (shortest) string that matches in full"""
    out = {}
    for di, d in enumerate(docs):
        for end in range(1, len(d) + 1):
            for st in range(end - 1, max(-1, end - 64), -1):
                if regex.fullmatch(d, st, end):
                    out.setdefault(d[st:end], []).append((di, st))
                    break
    return out


def _rows(occ):
    """the --count rows: one per distinct matched string, longest first, then by byte value (matchcmp, search_tool.cc:118)"""
    keys = sorted(occ, key=lambda s: (-len(s), s))
    return b"".join(b"% 4d \"%s\"\n" % (len(occ[s]), _alphatos(s)) for s in keys)


def test_search_cli_string_queries(fixtures, tmp_path, gpu_ok):
    """a query that is one string (simplify_query): --count / --matches print its row and the total (search_tool.cc:1075-1112
    prints the rows for either flag), the document list and --offsets follow print_matches"""
    fx = fixtures("eng2doc")
    docs = [d.tobytes() for d in fx.docs]
    infos = [os.path.basename(p).encode() for p in fx.doc_paths]
    for pattern in [b"the", b"and ", b"zzzzqq", docs[1][100:117], b"e. "]:
        q = testpl.x_escaped(pattern)                     # whitespace and punctuation have to be quoted in femto's language
        occ = _occurrences(docs, re.compile(re.escape(pattern), re.S))
        hits = sorted(occ.get(pattern, []))
        total = len(hits)
        for flag in ("--count", "--matches"):
            r = _run([flag, fx.index, q])
            assert r.returncode == 0 and r.stdout == _rows(occ) + b"% 4d total matches\n" % total, (pattern, flag, r.stdout, r.stderr)
        want_docs, want_offs = b"", b""
        for d in sorted({h[0] for h in hits}):
            want_docs += infos[d] + b"\n"
            want_offs += infos[d] + b"\n\t" + b"".join(b" %d" % o for dd, o in hits if dd == d) + b"\n"
        assert _run([fx.index, q]).stdout == want_docs
        out = str(tmp_path / "o.txt")
        assert _run(["--offsets", "--output", out, "--pattern", q, fx.index]).returncode == 0
        assert open(out, "rb").read() == want_offs
        assert _run(["--offsets", "--null", fx.index, q]).stdout == want_offs.replace(b"\n", b"\0")
        # the bytes themselves: --raw-pattern (string_node_new(rawpattern), search_tool.cc:716-718) and round 4's --literal
        if b"\0" not in pattern:
            assert _run(["--offsets", "--raw-pattern", pattern, fx.index]).stdout == want_offs
            assert _run(["--literal", "--offsets", fx.index, pattern]).stdout == want_offs
        rf = str(tmp_path / "raw.bin")
        open(rf, "wb").write(pattern)
        assert _run(["--offsets", "--raw-pattern-from", rf, fx.index]).stdout == want_offs
    # unescaped whitespace separates terms: "th e" is the string "the"
    assert _run(["--count", fx.index, "th e"]).stdout == _run(["--count", fx.index, "the"]).stdout
    # two indexes: the same string found in both is ONE row with the sum
    n1 = int(_run(["--count", fx.index, "the"]).stdout.split()[0])
    flat = str(tmp_path / "index.flat")                    # the flattened container is the same index
    femto_amd.flatten_index(fx.index, flat)
    r = _run(["--count", fx.index, flat, "the"])
    assert r.stdout == b"% 4d \"the\"\n% 4d total matches\n" % (2 * n1, 2 * n1)
    r = _run(["--count", "--by_index", fx.index, flat, "the"]).stdout.split(b"\n")
    assert r[0] == b"Results from " + fx.index.encode() and r[2] == b"Results from " + flat.encode() and r[1].startswith(b"% 4d [" % n1)
    # --max_results: the first chunk of rows (do_range_to_results_query, server.c:4754): rows first .. first + n - 1
    ix = femto_amd.Index(fx.index, device=0)
    first, last = ix.count([np.frombuffer(b"the", dtype=np.uint8).astype(np.uint16) + 5])
    assert last[0] - first[0] + 1 == n1 > 7
    rows7 = sorted(ix.resolve_location(int(o)) for o in ix.locate_range(int(first[0]), int(first[0]) + 6))
    ix.close()
    want = b""
    for d in sorted({h[0] for h in rows7}):
        want += infos[d] + b"\n\t" + b"".join(b" %d" % o for dd, o in rows7 if dd == d) + b"\n"
    assert _run(["--offsets", "--max_results", "7", fx.index, "the"]).stdout == want


def test_search_cli_regular_expressions(fixtures, tmp_path, gpu_ok):
    """femto_search's pattern is a regular expression: every distinct matched string is a --count row; documents and offsets
    are the union over the result ranges; --icase, --json, APPROX"""
    fx = fixtures("eng2doc")
    docs = [d.tobytes() for d in fx.docs]
    infos = [os.path.basename(p).encode() for p in fx.doc_paths]
    for q, py in [(rb"th[ae]", rb"th[ae]"), (rb"q.", rb"q."), (rb"\ (a|an|the)\ ", rb" (a|an|the) "), (rb"e\.\ [A-Z]", rb"e\. [A-Z]"),
                  (rb"ing{2}", rb"ing{2}"), (rb"(ab|cd)x", rb"(ab|cd)x")]:
        occ = _occurrences(docs, re.compile(py, re.S))
        total = sum(len(v) for v in occ.values())
        r = _run(["--count", fx.index, q])
        assert r.returncode == 0 and r.stdout == _rows(occ) + b"% 4d total matches\n" % total, (q, r.stdout[:400], r.stderr)
        hits = sorted(set(h for v in occ.values() for h in v))
        want = b""
        for d in sorted({h[0] for h in hits}):
            want += infos[d] + b"\n\t" + b"".join(b" %d" % o for dd, o in hits if dd == d) + b"\n"
        assert _run(["--offsets", fx.index, q]).stdout == want, q
        assert _run([fx.index, q]).stdout == b"".join(infos[d] + b"\n" for d in sorted({h[0] for h in hits})), q
    # streamline_query (query_planning.c): optional parts at the ends are dropped -- "x*the+" is searched as "the"
    assert _run(["--count", fx.index, "x*the+"]).stdout == _run(["--count", fx.index, "the"]).stdout
    # --icase (icase_ast): every letter in both cases
    occ = _occurrences(docs, re.compile(rb"[Tt][Hh][Ee]", re.S))
    r = _run(["--count", "--icase", fx.index, "The"])
    assert r.stdout == _rows(occ) + b"% 4d total matches\n" % sum(len(v) for v in occ.values())
    # APPROX 1: the strings within one edit are the rows (a string whose range lies inside another result's -- "whichx" inside
    # "which" -- is dropped by regexp_result_list_sort, so the total covers the exact matches without naming them)
    word = re.search(rb"[a-z]{6}", docs[0][500:]).group(0).decode()       # six letters that do occur
    exact = int(_run(["--count", fx.index, word]).stdout.split()[0])
    r = _run(["--count", fx.index, "APPROX 1 " + word])
    assert r.returncode == 0 and exact > 0 and int(r.stdout.split(b"\n")[-2].split()[0]) >= exact and r.stdout.count(b"\n") >= 2
    o = femto_amd.Index(fx.index, device=0)
    f, l, m, c = o.regexp_search(word.encode(), approx=(1, 1, 1, 1))
    o.close()
    assert int(r.stdout.split(b"\n")[-2].split()[0]) == int((l - f + 1).sum()) and r.stdout.count(b"\n") - 1 <= len(f)
    # --json: one object, the query echoed as ast_to_string prints it, rows as [string, count] / [[info], [offsets]]
    occ = _occurrences(docs, re.compile(rb"th[ae]", re.S))
    j = json.loads(_run(["--count", "--json", fx.index, "th[ae]"]).stdout)
    assert j["pattern"] == "th[ae]" and j["total"] == sum(len(v) for v in occ.values())
    assert j["results"] == [[s.decode(), len(occ[s])] for s in sorted(occ, key=lambda s: (-len(s), s))]
    j = json.loads(_run(["--offsets", "--json", fx.index, "zzzzqq|" + "the"]).stdout)
    hits = sorted(_occurrences(docs, re.compile(rb"the", re.S))[b"the"])
    assert j["pattern"] == '( "zzzzq"q| "the")'      # (a word before punctuation leaves its last letter behind, posix.flex.l:293)
    assert j["results"] == [[[infos[d].decode()], [o for dd, o in hits if dd == d]] for d in sorted({h[0] for h in hits})]
    # what femto_search cannot do here is refused by name, not misread
    for args, msg in ((["--grep", fx.index, "the"], b"--grep is not supported"), (["--suggest", "--count", fx.index, "the"], b"--suggest is not supported"),
                      ([fx.index, "black AND sheep"], b"Could not parse pattern"), ([fx.index, "(the"], b"Could not parse pattern"),
                      (["--count", str(tmp_path / "nowhere"), "the"], b"Could not open index")):
        r = _run(args)
        assert r.returncode != 0 and msg in r.stderr + r.stdout, (args, r.stdout, r.stderr)


def test_search_cli_passes_the_reference_end_to_end_test(tmp_path, gpu_ok):
    """The reference's own end-to-end test of femto_search, src/test/test.pl, with femto_amd_search in femto_search's place and
    the SAME command lines (test.pl:497-507: `--offsets --output <file> <index> '<query>'`, then without --offsets): the same
    sixteen fixed documents plus twenty random ones, indexed with its parameters under the names it gives them, queried with
    its literal queries -- escaped as its x_escaped() escapes them, i.e. as regular expressions -- and with its generated
    regular-expression queries (add_query's nineteen templates over the first bytes of eighteen documents), the output read
    back by its own parser (parseresults) and compared with what its Perl search expects (tests/testpl.py restates all of
    that; tests/test_query_language.py runs the WHOLE query set against the prepared automata on the CPU).  Every run of the
    tool opens the index on the GPU, so a spread of ~110 of the ~500 queries keeps this to a few minutes."""
    docs, queries = testpl.make_docs_and_queries(1)
    indir = tmp_path / "input"
    indir.mkdir()
    infos = [str(indir / ("%03d" % i)) for i in range(len(docs))]
    index = str(tmp_path / "index")
    femto_amd.build_index(index, [np.frombuffer(d, dtype=np.uint8) for d in docs], params=testpl.INDEX_PARAMS, infos=infos, device=0)
    out = str(tmp_path / "results")
    regex = [q for q in queries if femto_amd.Nfa.from_query(q)[1] is None]
    literal = [q for q in queries if q not in set(regex)]
    keep = [q for q in literal if q in (rb"\x09", rb"\x0a", rb"\x00\x00", rb"\x00\x01", rb"\x00\x01\x00", rb"\x01\x00")]
    sample = keep + literal[::max(1, len(literal) // 30)] + regex[::max(1, len(regex) // 75)]
    assert len(keep) == 6 and len(regex) > 80
    for q in sample:
        expected = testpl.expected_results(docs, queries[q])
        r = _run(["--offsets", "--output", out, index, q])
        assert r.returncode == 0, (q, r.stderr)
        assert testpl.parseresults(open(out, "rb").read()) == expected, q
        r = _run(["--output", out, index, q])
        assert r.returncode == 0, (q, r.stderr)
        assert testpl.parseresults(open(out, "rb").read()) == [(i, []) for i, _ in expected], q

#!/usr/bin/env python3
"""Golden vectors of regular-expression / automaton search from the GENUINE do_regexp_query (container only).

    python tests/golden/make_regexp_golden.py

For the committed fixture indexes (built by the reference, make_golden.py) it feeds automata to the reference through
setup_regexp_query_take_nfa (src/main/server.h:838; oracle/ref_tool.c `regexp_nfa`) and stores, per fixture,
`<name>_regexp.npz`: the automata (the flat form of nfa_description_t, src/main/nfa.h:62-88) and the reference's sorted
result lists {first, last, match_len, cost}.  The automata are
  * the position automata femto_amd_regexp_compile builds from pattern texts (exact and APPROX settings) -- the pattern
    text is stored next to the automaton so that the construction itself is pinned too,
  * hand-made random automata (acyclic and cyclic, exact and with error costs), independent of our parser.
(The reference's own regex FRONT END needs flex/bison and is not built; the search below it is what is pinned.)
Only data is committed: automata and expected results.
"""
import os
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import femto_amd  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

PATTERNS = {
    "acgt48k": [rb"ACGTACG", rb"AC(GT|TG)+A", rb"G[AC]T[^A]GG", rb"TTT.TTT", rb"(ACG|TGCA)A?C", rb"GATTACA|TACAGAT", rb"A*", rb"",
                rb"(AC)*G", rb"[AC][AC][GT][GT]ACG", rb"T(A|C|G)?T+GCA", rb"CCCCCC.?.?G", rb"(A|C)(C|G)(G|T)(T|A)(A|C)(C|G)TT",
                rb"ACGT.*TGCA", rb"N", rb"AC\x47T"],
    "eng2doc": [rb"the", rb"th[aeiou]+", rb"(and|or)\ [a-z]+", rb"[A-Z][a-z]+ing", rb"e\.\ ", rb"q.", rb"wor(d|k)s?", rb"[0-9]+",
                rb"\ (a|an|the)\ ", rb"\n.", rb"ing[,.;]", rb'"of the"', rb"[^a-z ]"],
    "bytes256": [rb"\x00.", rb"[\x80-\xff][\x00-\x10]", rb"\xfe|\xff\xff?", rb"[\x00-\xff]\x7f", rb"\x01\x02?\x03"],
    "runs3doc": [rb"a+b", rb"(ab)+", rb"aaaa*", rb"b.b", rb"[ab]*c"],
    "chunks2doc": [rb"the", rb"[a-z]+\.", rb"(in|on|at)\ "],
}
APPROX = {
    "acgt48k": [(rb"ACGTACGTTGCA", (1, 1, 1, 1)), (rb"GATTACAGATTA", (1, 1, 1, 1)), (rb"GATTACAGATTA", (2, 1, 1, 1)),
                (rb"AC(GT|TG)+ACCA", (1, 1, 1, 1)), (rb"GATTACAGATTA", (1, 2, 1, 2)), (rb"GATTACAGATTA", (2, 1, 2, 1)),
                (rb"GATTACAGATTACCA", (2, 1, 1, 1)), (rb"CC[AG]TTGACCAT", (2, 2, 1, 1)), (rb"TGCATGCATG", (2, 1, 2, 2))],
    "eng2doc": [(rb"because", (1, 1, 1, 1)), (rb"government", (2, 1, 1, 1)), (rb"th(is|at)\ ", (1, 1, 1, 1)), (rb"qu[aeiou]ck", (1, 1, 2, 1))],
    "bytes256": [(rb"\x10\x11\x12\x13", (1, 1, 1, 1))],
}


def random_nfa(rng, alpha, approx, cyclic):
    """every non-final node has a transition (the reference waits for ever on a pending range whose states can read
    nothing, server.c:1954-1990); insert_cost <= subst_cost keeps all-dead children out for the same reason"""
    n = int(rng.integers(2, 10))
    ts, tc, td = [0], [], []
    for i in range(n):
        k = int(rng.integers(1, 5)) if i < n - 1 else (int(rng.integers(0, 3)) if cyclic else 0)
        for _ in range(k):
            tc.append(int(rng.choice(alpha)))
            td.append(int(rng.integers(0, n)) if cyclic else int(rng.integers(i + 1, n)))
        ts.append(len(tc))
    st = (rng.random(n) < 0.3).astype(np.uint8)
    st[0] = 1
    fi = (rng.random(n) < 0.2).astype(np.uint8)
    fi[n - 1] = 1
    if approx:
        b = int(rng.integers(2, 4))
        c = [int(rng.integers(1, 3)) for _ in range(3)]
        c[0] = max(c[0], (b + 2) // 3)
        c[2] = max(c[2], (b + 2) // 3)
        c[0] = max(c[0], c[2])
        se = (b, c[0], c[1], c[2])
    else:
        se = (1, 1, 1, 1)
    return femto_amd.Nfa(ts, tc, td, st, fi, se)


def run_ref(index, nfas, cyclic, td):
    """one reference call for the patterns and the acyclic automata; one call per cyclic automaton with a timeout (a cyclic
    automaton may explore until MAX_REGEXP_ITERATIONS: the slow ones are dropped)"""
    kept, res = list(nfas), po.ref_regexp_nfa(index, nfas, td, timeout=600)
    for a in cyclic:
        try:
            r = po.ref_regexp_nfa(index, [a], td, timeout=5)
        except subprocess.TimeoutExpired:
            continue
        kept.append(a)
        res.append(r[0])
    return kept, res


def main():
    for name, pats in PATTERNS.items():
        with tempfile.TemporaryDirectory() as td:
            with tarfile.open(os.path.join(OUT, name + ".tar.gz")) as tf:
                tf.extractall(td)
            index = os.path.join(td, "index")
            docs = [np.fromfile(os.path.join(td, f), dtype=np.uint8) for f in sorted(os.listdir(td)) if f.startswith("doc")]
            text_chars = np.unique(np.concatenate(docs)).astype(np.int32) + 5
            labelled = [(femto_amd.Nfa.from_regex(p or b"''"), p, (0, 1, 1, 1)) for p in pats]      # '' is femto's spelling of the empty pattern
            labelled += [(femto_amd.Nfa.from_regex(p, k), p, k) for p, k in APPROX.get(name, [])]
            rng = np.random.Generator(np.random.PCG64(sum(name.encode()) + 17))
            # characters of the text (twice: most transitions can be followed), SEOF, one character the text lacks
            lacks = [c for c in range(5, 261) if c not in set(text_chars.tolist())][:1]
            alpha = np.concatenate([text_chars, text_chars, np.array([2] + lacks, dtype=np.int32)])
            if len(text_chars) > 16:      # byte texts: a random character seldom follows another -- draw from a small set
                alpha = np.concatenate([text_chars[:6], text_chars[:6], np.array([2] + lacks, dtype=np.int32)])
            hand = [random_nfa(rng, alpha, approx=False, cyclic=False) for _ in range(40)]
            hand += [random_nfa(rng, alpha, approx=True, cyclic=False) for _ in range(30)]
            cyc = [random_nfa(rng, alpha, approx=False, cyclic=True) for _ in range(16)]
            cyc += [random_nfa(rng, alpha, approx=True, cyclic=True) for _ in range(8)]
            nfas = [a for a, _, _ in labelled] + hand
            kept, res = run_ref(index, nfas, cyc, td)
            hand += cyc
            nfas += cyc
            keep_ids = {id(a) for a in kept}
            regex = [p for a, p, _ in labelled if id(a) in keep_ids] + [b""] * sum(1 for a in hand if id(a) in keep_ids)
            from_regex = [1 for a, _, _ in labelled if id(a) in keep_ids] + [0] * sum(1 for a in hand if id(a) in keep_ids)
            approx = [k for a, _, k in labelled if id(a) in keep_ids] + [(0, 0, 0, 0)] * sum(1 for a in hand if id(a) in keep_ids)
            n = len(kept)
            gold = dict(
                n=np.int64(n),
                num_nodes=np.array([a.num_nodes for a in kept], dtype=np.int32),
                settings=np.array([a.settings for a in kept], dtype=np.int32),
                trans_start=np.concatenate([a.trans_start for a in kept]),
                trans_char=np.concatenate([a.trans_char for a in kept]),
                trans_dest=np.concatenate([a.trans_dest for a in kept]),
                is_start=np.concatenate([a.is_start for a in kept]),
                is_final=np.concatenate([a.is_final for a in kept]),
                regex=np.array(regex, dtype="S64"),
                from_regex=np.array(from_regex, dtype=np.uint8),
                approx=np.array(approx, dtype=np.int32),
                err_code=np.array([r[0] for r in res], dtype=np.int32),
                res_count=np.array([len(r[1]) for r in res], dtype=np.int64),
                res_first=np.concatenate([r[1] for r in res]),
                res_last=np.concatenate([r[2] for r in res]),
                res_len=np.concatenate([r[3] for r in res]),
                res_cost=np.concatenate([r[4] for r in res]),
            )
            np.savez_compressed(os.path.join(OUT, name + "_regexp.npz"), **gold)
            print(name, n, "automata (of", len(nfas), "),", int(gold["res_count"].sum()), "results,",
                  int((gold["err_code"] != 0).sum()), "errors,", os.path.getsize(os.path.join(OUT, name + "_regexp.npz")), "bytes")


if __name__ == "__main__":
    main()

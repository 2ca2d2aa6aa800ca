"""The reference's end-to-end test of femto_search, src/test/test.pl, restated as data generators (no Perl source is kept):
the documents it indexes, the queries it asks -- every literal escaped the way its x_escaped() escapes it, and the
regular-expression templates of its add_query() -- and the answers it expects, which it finds with Perl's regex engine and
this file finds with Python's `re` (the constructs test.pl uses -- literals, `.` with /s, classes, groups, alternation and
the non-greedy repeats -- mean the same in both).

test.pl seeds Perl's rand(); Python cannot replay that stream, so the random documents and queries are drawn from the same
distributions (test.pl:66-88) with numpy's PCG64 -- the fixed documents and every query TEMPLATE are test.pl's own."""
import re

import numpy as np

FIXED_DOCS = [b"a", b"aa", b"aab", b"aac", b"bb", b"test", b"fun", b"\x00\x00", b"\x00\x01\x00", b"bannana", b"seeresses", b"equal",
              b"un", b"undo", b"bbababcc", bytes(range(256))]                      # test.pl:57-60
INDEX_PARAMS = "mark_period=20 bucket_size=1048576 block_size=16777216"          # test.pl:25-28 (chunk_size=64: see make_index)
NUM_ADDITIONAL_DOCS, MIN_ADD_DOC, MAX_ADD_DOC = 20, 1, 500                        # test.pl:31-35
NUM_RANDOM_QUERIES, MIN_RAND_QUERY, MAX_RAND_QUERY = 50, 1, 16                   # test.pl:37-39
NUM_DOCS_REGEXPD = 18                                                             # test.pl:42
TEMPLATES = ["A.", "B.D", "AB.DE", "A+B", "(ABC)+D", "A?B", "AB?C", "AB*C", "A{2}D", "A{2,}D", "A{2,3}E", "[AB]", "[^AB]",
             "[A-B]", "[^A-B]", "AB|CD", "(AB|CD)+E", "AB(CD|EF)CD", "AB(CD|EF)+CD"]     # test.pl:344-365

_PUNCT = set(b"!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")


def _is_word(c):        # Perl's \w on a byte: [A-Za-z0-9_]
    return c == 0x5f or 0x30 <= c <= 0x39 or 0x41 <= c <= 0x5a or 0x61 <= c <= 0x7a


def x_escaped(s):
    """test.pl:89-111: punctuation (but not the single quote: the query travels inside '...' through the shell) gets a
    backslash, every other non-word byte becomes \\xNN"""
    out = b""
    for c in s:
        if _is_word(c):
            out += bytes([c])
        elif c in _PUNCT and c != 0x27:
            out += b"\\" + bytes([c])
        else:
            out += b"\\x%02x" % c
    return out


def make_docs_and_queries(seed=1):
    """(docs, {femto query text: compiled Python regex}) in test.pl's shape"""
    rng = np.random.Generator(np.random.PCG64(seed))

    def randstr(n):                         # test.pl:66-88: any bytes / chr(40..125) / lowercase letters
        t = int(rng.integers(1, 4))
        lo, hi = {1: (0, 256), 2: (40, 126), 3: (97, 123)}[t]
        return bytes(rng.integers(lo, hi, n).astype(np.uint8).tolist())

    docs = list(FIXED_DOCS) + [randstr(int(rng.integers(MIN_ADD_DOC, MAX_ADD_DOC))) for _ in range(NUM_ADDITIONAL_DOCS)]
    literal = {b"\t", b"\n"}                                                       # test.pl:253-254
    for d in docs:                                                                 # :263-270
        for a, n in ((0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4)):
            literal.add(d[a:a + n])
    for _ in range(NUM_RANDOM_QUERIES):                                            # :273-276
        literal.add(randstr(int(rng.integers(MIN_RAND_QUERY, MAX_RAND_QUERY))))
    literal.discard(b"")
    queries = {}
    for q in sorted(literal):                                                      # :286-300: quotemeta for Perl, x_escaped for femto
        queries[x_escaped(q)] = re.compile(re.escape(q), re.S)
    for doc in docs[:NUM_DOCS_REGEXPD]:                                            # :332-366
        subs = [(re.escape(bytes([0x61 + k])), bytes([0x61 + k]), 0x61 + k) for k in range(7)]
        for k in range(min(7, len(doc))):
            subs[k] = (re.escape(doc[k:k + 1]), x_escaped(doc[k:k + 1]), doc[k])
        for t in TEMPLATES:
            if t in ("[A-B]", "[^A-B]") and not subs[0][2] < subs[1][2]:
                continue
            py, fe = b"", b""
            for ch in t.encode():                                                  # add_query, :302-327
                k = ch - 0x41
                if 0 <= k < 7:
                    py += subs[k][0]
                    fe += subs[k][1]
                else:
                    py += {0x2a: b"*?", 0x2b: b"+?", 0x3f: b"??", 0x7d: b"}?"}.get(ch, bytes([ch]))
                    fe += bytes([ch])
            queries[fe] = re.compile(py, re.S)
    return docs, queries


def expected_results(docs, regex):
    """test.pl:384-414: per document the offsets j at which the (non-greedy) pattern matches, where a match that ends where the
    match found at the previous offset ended REPLACES it -- [(doc, [offsets])] for documents with a match"""
    out = []
    for i, d in enumerate(docs):
        offs, lastend = [], -1
        for j in range(len(d)):
            m = regex.match(d, j)
            if m:
                if m.end() == lastend:
                    offs[-1] = j
                else:
                    offs.append(j)
                    lastend = m.end()
        if offs:
            out.append((i, offs))
    return out


def parseresults(data):
    """test.pl:115-145: a line starting with TAB holds the offsets of the document named on the line before; the document id
    is the basename of its path"""
    import os
    ret = []
    for line in data.split(b"\n")[:-1]:
        if line[:1] == b"\t":
            offs = sorted(int(x) for x in line.split())
            doc_only = ret.pop()        # "remove document-only"
            ret.append((doc_only[0], offs))
        else:
            ret.append((int(os.path.basename(line)), []))
    return sorted(ret)

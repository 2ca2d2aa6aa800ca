"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/femto_oracle.h).  The product package (femto_amd/) never does.

Two back ends:
  * Oracle      -- oracle/_ref/libfemto_oracle.so, our plain-C restatement (always available
                   after `make -C oracle oracle`).
  * ref_tool()  -- oracle/_ref/ref_tool, the genuine reference compiled from /root/reference by
                   oracle/Makefile (prebuilt binary travels to the GPU box; absent -> None).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
ORACLE_SO = os.path.join(REF_DIR, "libfemto_oracle.so")
REF_TOOL = os.path.join(REF_DIR, "ref_tool")
REF_TOOL_AMD = os.path.join(REF_DIR, "ref_tool_amd")       # ref_tool linked with integration/femto_amd_shim.c
INDEX_TEST_AMD = os.path.join(REF_DIR, "index_test_amd")   # the reference's index_test.c linked with the shim
FPAT_MAGIC = 0x54415046


def build(oracle_only=False):
    """Compile the oracle (and, when /root/reference exists, the reference) -- building the
    checker is not using it."""
    target = ["oracle"] if oracle_only else []
    subprocess.run(["make", "-C", HERE, "-j8", "-s"] + target, check=True)
    # the reference's own callers linked against integration/femto_amd_shim.c (needs the built product library)
    if not oracle_only and os.path.exists(os.path.join(os.path.dirname(HERE), "femto_amd", "libfemto_amd.so")):
        subprocess.run(["make", "-C", HERE, "-j8", "-s", "shim"], check=True)


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n_rank", "n_occ", "n_mark", "s_bytes", "n_rle", "n_lit", "n_gamma", "n_lf")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(oracle_only=True)
        L = C.CDLL(ORACLE_SO)
        L.fo_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.fo_close.argtypes = [C.c_void_p]
        L.fo_total_length.restype = C.c_int64
        L.fo_total_length.argtypes = [C.c_void_p]
        L.fo_num_blocks.restype = C.c_int64
        L.fo_num_blocks.argtypes = [C.c_void_p]
        L.fo_num_documents.restype = C.c_int64
        L.fo_num_documents.argtypes = [C.c_void_p]
        L.fo_param.argtypes = [C.c_void_p, C.c_int]
        L.fo_get_C.restype = C.c_int64
        L.fo_get_C.argtypes = [C.c_void_p, C.c_int]
        L.fo_get_block_occs.restype = C.c_int64
        L.fo_get_block_occs.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        L.fo_bseq_rank.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.fo_block_request.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p]
        L.fo_count.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.fo_locate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.fo_forward_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p]
        L.fo_back_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p]
        L.fo_bseq_rank_all.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.fo_decode_gamma.argtypes = [C.c_uint64, C.POINTER(C.c_uint)]
        L.fo_decode_varbyte.argtypes = [C.c_char_p, C.POINTER(C.c_uint)]
        L.fo_resolve_location.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.fo_wtree_occs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.fo_wtree_rank.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.fo_nfa_search.restype = C.c_int64
        L.fo_nfa_search.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int64, C.c_int64] + [C.c_void_p] * 4
        _lib = L
    return _lib


def _flat(patterns):
    """list of uint16 arrays -> (plen int32[n], flat uint16[sum], starts int64[n])"""
    plen = np.array([len(p) for p in patterns], dtype=np.int32)
    starts = np.zeros(len(patterns), dtype=np.int64)
    if len(patterns):
        starts[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    flat = (np.concatenate([np.asarray(p, dtype=np.uint16) for p in patterns])
            if len(patterns) and plen.sum() else np.zeros(0, dtype=np.uint16))
    return plen, np.ascontiguousarray(flat), starts


class Oracle:
    """The C restatement, opened on a femto index (directory or flattened file)."""

    def __init__(self, path):
        self.h = C.c_void_p()
        rc = lib().fo_open(os.fsencode(path), C.byref(self.h))
        if rc:
            raise RuntimeError(f"fo_open({path}) -> err_code {rc}")
        self.total_length = lib().fo_total_length(self.h)
        self.num_blocks = lib().fo_num_blocks(self.h)
        self.block_size = lib().fo_param(self.h, 0)
        self.b_size = lib().fo_param(self.h, 1)
        self.mark_period = lib().fo_param(self.h, 2)

    def close(self):
        if self.h:
            lib().fo_close(self.h)
            self.h = None

    def C(self, ch):
        return lib().fo_get_C(self.h, ch)

    def block_occs(self, ch, block):
        return lib().fo_get_block_occs(self.h, ch, block)

    def block_request(self, row, kind, ch=0x1FF):
        """kind bits 1 CHAR 2 OCCS 4 LOCATION on a GLOBAL row; returns (ch, occs_in_block, offset)"""
        blk = row // self.block_size
        c = C.c_int(ch)
        occ = C.c_int(0)
        off = C.c_int64(-1)
        rc = lib().fo_block_request(self.h, blk, kind, int(row - blk * self.block_size),
                                    C.byref(c), C.byref(occ), C.byref(off), None)
        if rc:
            raise RuntimeError(f"fo_block_request -> {rc}")
        return c.value, occ.value, off.value

    def forward_step(self, row):
        """do_forward_query: (chr, new_row, offset)"""
        nr, ch, off = C.c_int64(), C.c_int(), C.c_int64()
        rc = lib().fo_forward_step(self.h, row, C.byref(nr), C.byref(ch), C.byref(off), None)
        if rc:
            raise RuntimeError(f"fo_forward_step -> {rc}")
        return ch.value, nr.value, off.value

    def back_step(self, row):
        """do_back_query: (chr, new_row, offset)"""
        nr, ch, off = C.c_int64(), C.c_int(), C.c_int64()
        rc = lib().fo_back_step(self.h, row, C.byref(nr), C.byref(ch), C.byref(off), None)
        if rc:
            raise RuntimeError(f"fo_back_step -> {rc}")
        return ch.value, nr.value, off.value

    def count_flat(self, plen, flat, starts, threads=1, counters=None):
        n = len(plen)
        first = np.zeros(n, dtype=np.int64)
        last = np.zeros(n, dtype=np.int64)
        rc = lib().fo_count(self.h, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data,
                            first.ctypes.data, last.ctypes.data, threads,
                            C.byref(counters) if counters is not None else None)
        if rc:
            raise RuntimeError(f"fo_count -> {rc}")
        return first, last

    def count(self, patterns, threads=1, counters=None):
        return self.count_flat(*_flat(patterns), threads=threads, counters=counters)

    def locate_flat(self, plen, flat, starts, max_occs, threads=1, counters=None):
        n = len(plen)
        noccs = np.zeros(n, dtype=np.int32)
        rc = lib().fo_locate(self.h, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data,
                             max_occs, noccs.ctypes.data, None, threads, None)
        if rc:
            raise RuntimeError(f"fo_locate -> {rc}")
        offs = np.zeros(int(noccs.sum()) + 1, dtype=np.int64)
        rc = lib().fo_locate(self.h, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data,
                             max_occs, noccs.ctypes.data, offs.ctypes.data, threads,
                             C.byref(counters) if counters is not None else None)
        if rc:
            raise RuntimeError(f"fo_locate -> {rc}")
        return noccs, offs[:-1]

    def locate(self, patterns, max_occs, threads=1, counters=None):
        return self.locate_flat(*_flat(patterns), max_occs, threads=threads, counters=counters)

    def nfa_search(self, nfa, max_iterations=1000000, cap=1 << 20):
        """do_regexp_query restated (fo_nfa_search): (err_code, first, last, match_len, cost) for one automaton (an object
        with trans_start / trans_char / trans_dest int32, is_start / is_final uint8 and settings, e.g. femto_amd.Nfa)"""
        first, last = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
        mlen, cost = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
        st = np.array(nfa.settings, dtype=np.int32)
        n = lib().fo_nfa_search(self.h, nfa.num_nodes, nfa.trans_start.ctypes.data, nfa.trans_char.ctypes.data, nfa.trans_dest.ctypes.data,
                                nfa.is_start.ctypes.data, nfa.is_final.ctypes.data, st.ctypes.data, max_iterations, cap,
                                first.ctypes.data, last.ctypes.data, mlen.ctypes.data, cost.ctypes.data)
        if n < 0:
            return -n, first[:0], last[:0], mlen[:0], cost[:0]
        assert n <= cap
        return 0, first[:n], last[:n], mlen[:n], cost[:n]


def bseq_rank(image, index):
    """(occs0, occs1, bit) of one encoded binary sequence at 1-based index"""
    buf = (C.c_ubyte * (len(image) + 64)).from_buffer_copy(bytes(image) + b"\0" * 64)
    occs = (C.c_int * 2)()
    bit = C.c_int()
    lib().fo_bseq_rank(buf, index, occs, C.byref(bit), None)
    return occs[0], occs[1], bit.value


def bseq_rank_all(image, nbits):
    buf = (C.c_ubyte * (len(image) + 64)).from_buffer_copy(bytes(image) + b"\0" * 64)
    out = np.zeros((nbits, 3), dtype=np.int64)
    lib().fo_bseq_rank_all(buf, nbits, C.c_void_p(out.ctypes.data))
    return out


# ----------------------------------------------------------------------------- reference binary

def have_ref():
    return os.path.exists(REF_TOOL)


def write_fpat(path, patterns):
    plen, flat, _ = _flat(patterns)
    with open(path, "wb") as f:
        f.write(np.array([FPAT_MAGIC, len(plen)], dtype=np.uint32).tobytes())
        f.write(plen.tobytes())
        f.write(flat.tobytes())


def write_fpat_flat(path, plen, flat):
    with open(path, "wb") as f:
        f.write(np.array([FPAT_MAGIC, len(plen)], dtype=np.uint32).tobytes())
        f.write(np.ascontiguousarray(plen, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(flat, dtype=np.uint16).tobytes())


def ref_tool(*args, capture=True, timeout=None):
    r = subprocess.run([REF_TOOL] + [str(a) for a in args], check=True,
                       stdout=subprocess.PIPE if capture else None, timeout=timeout)
    return r.stdout.decode() if capture else ""


def ref_build(index_dir, params, doc_paths):
    return json.loads(ref_tool("build", index_dir, params or "-", *doc_paths))


def ref_dump(index_path, tmp):
    ref_tool("dump", index_path, tmp)
    raw = open(tmp, "rb").read()
    n, nblocks = np.frombuffer(raw, dtype=np.int64, count=2)
    o = 16
    Carr = np.frombuffer(raw, dtype=np.int64, count=262, offset=o); o += 262 * 8
    bo = np.frombuffer(raw, dtype=np.int64, count=261 * nblocks, offset=o).reshape(261, nblocks); o += 261 * nblocks * 8
    L = np.frombuffer(raw, dtype=np.uint16, count=n, offset=o); o += 2 * n
    occ = np.frombuffer(raw, dtype=np.int32, count=n, offset=o); o += 4 * n
    off = np.frombuffer(raw, dtype=np.int64, count=n, offset=o)
    return dict(n=int(n), nblocks=int(nblocks), C=Carr.copy(), block_occs=bo.copy(), L=L.copy(),
                occ=occ.copy(), off=off.copy())


def ref_forward(index_path, tmp):
    ref_tool("forward", index_path, tmp)
    raw = open(tmp, "rb").read()
    n = len(raw) // 18
    ch = np.frombuffer(raw, dtype=np.uint16, count=n).copy()
    nr = np.frombuffer(raw, dtype=np.int64, count=n, offset=2 * n).copy()
    off = np.frombuffer(raw, dtype=np.int64, count=n, offset=10 * n).copy()
    return ch, nr, off


def ref_count(index_path, patterns, tmpdir):
    pf, of = os.path.join(tmpdir, "p.fpat"), os.path.join(tmpdir, "count.bin")
    write_fpat(pf, patterns)
    ref_tool("count", index_path, pf, of)
    a = np.fromfile(of, dtype=np.int64)
    n = len(patterns)
    return a[:n].copy(), a[n:2 * n].copy()


def ref_locate(index_path, patterns, max_occs, tmpdir):
    pf, of = os.path.join(tmpdir, "p.fpat"), os.path.join(tmpdir, "loc.bin")
    write_fpat(pf, patterns)
    ref_tool("locate", index_path, pf, max_occs, of)
    raw = open(of, "rb").read()
    n = len(patterns)
    noccs = np.frombuffer(raw, dtype=np.int32, count=n).copy()
    offs = np.frombuffer(raw, dtype=np.int64, offset=4 * n, count=int(noccs.sum())).copy()
    return noccs, offs


FNFA_MAGIC = 0x41464E46


def write_fnfa(path, nfas):
    """the automaton file ref_tool regexp_nfa reads (oracle/ref_tool.c)"""
    with open(path, "wb") as f:
        f.write(np.array([FNFA_MAGIC, len(nfas)], dtype=np.uint32).tobytes())
        for a in nfas:
            f.write(np.array([a.num_nodes, len(a.trans_char)] + list(a.settings), dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(a.trans_start, dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(a.trans_char, dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(a.trans_dest, dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(a.is_start, dtype=np.uint8).tobytes())
            f.write(np.ascontiguousarray(a.is_final, dtype=np.uint8).tobytes())


last_regexp_query_s = None


def ref_regexp_nfa(index_path, nfas, tmpdir, timeout=600):
    """the GENUINE do_regexp_query (setup_regexp_query_take_nfa, src/main/server.h:838) on hand-fed automata:
    one (err_code, first, last, match_len, cost) per automaton, in the order of the reference's sorted result list"""
    nf, of = os.path.join(tmpdir, "q.fnfa"), os.path.join(tmpdir, "regexp.bin")
    write_fnfa(nf, nfas)
    # (do_regexp_query never returns when a pending range's alive states can read NOTHING: it then schedules zero
    # requests and waits for them, server.c:1954-1990 -- hence the timeout; automata from a pattern have no such state)
    global last_regexp_query_s
    txt = ref_tool("regexp_nfa", index_path, nf, of, timeout=timeout)
    try:      # wall time inside femto_run_query only (process start, index open, result writing excluded)
        last_regexp_query_s = json.loads(txt.strip().splitlines()[-1])["query_s"]
    except Exception:      # noqa: BLE001
        last_regexp_query_s = None
    raw = open(of, "rb").read()
    out, o = [], 0
    rec = np.dtype([("first", "<i8"), ("last", "<i8"), ("len", "<i4"), ("cost", "<i4")])
    for _ in nfas:
        code, n = np.frombuffer(raw, dtype=np.int32, count=2, offset=o)
        o += 8
        r = np.frombuffer(raw, dtype=rec, count=int(n), offset=o)
        o += int(n) * rec.itemsize
        out.append((int(code), r["first"].copy(), r["last"].copy(), r["len"].copy(), r["cost"].copy()))
    return out

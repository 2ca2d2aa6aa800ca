"""femto_amd -- MI355X-native FM-index query engine reading femto (femto-dev/femto) index files.

This package is a thin ctypes binding over the C ABI in include/femto_amd.h (the product is the
shared library femto_amd/libfemto_amd.so: C++ host + hand-written HIP kernels for gfx950).  There
is NO CPU fallback: without the built library or without a HIP device every query call raises.

Host-side mirror of the reference's batch interface (src/main/femto_internal.h):
    Index.count(...)   <-> parallel_count   (src/main/femto.c:275)
    Index.locate(...)  <-> parallel_locate  (src/main/femto.c:331)
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

__all__ = ["Index", "FemtoAmdError", "lib", "ALPHA_SIZE", "CHARACTER_OFFSET", "Info"]

ALPHA_SIZE = 261
CHARACTER_OFFSET = 5
ERR_NAMES = {0: "NOERR", 1: "MEM", 2: "IO", 3: "PARAM", 4: "FORMAT", 5: "BZ_DATA", 6: "INVALID",
             8: "MISSING", 10: "FULL", 11: "OVERWORKED", 12: "UNKNOWN"}
ERR_PARAM, ERR_INVALID, ERR_FULL, ERR_OVERWORKED = 3, 6, 10, 11
BUDGET_ALL = -2       # femto_amd_options_t::hbm_budget_bytes: whatever is free on the device (the default is a bound, include/femto_amd.h)


class FemtoAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"femto_amd error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [("total_length", C.c_int64), ("number_of_blocks", C.c_int64), ("number_of_documents", C.c_int64),
                ("block_size", C.c_int32), ("bucket_size", C.c_int32), ("mark_period", C.c_int32),
                ("chunk_size", C.c_int32), ("text_size_bits", C.c_int32), ("total_buckets", C.c_int64),
                ("image_bytes", C.c_int64), ("table_bytes", C.c_int64)]


class Options(C.Structure):
    """femto_amd_options_t (include/femto_amd.h): every field -1 = auto after femto_amd_options_init"""
    _fields_ = [("struct_size", C.c_uint32), ("rank_mode", C.c_int32), ("hbm_budget_bytes", C.c_int64), ("packed_lines", C.c_int32),
                ("two_level_lines", C.c_int32), ("char_rank_lines", C.c_int32), ("text", C.c_int32), ("dense_arrays", C.c_int32),
                ("mark_every", C.c_int32), ("level_table", C.c_int32), ("level_table_syms", C.c_int32), ("level_table_bytes", C.c_int64),
                ("context_table", C.c_int32), ("context_syms", C.c_int32), ("context2_table", C.c_int32), ("context2_syms", C.c_int32),
                ("context2_bytes", C.c_int64), ("tail_min", C.c_int32), ("tail_ones", C.c_int32), ("tail_rows", C.c_int32),
                ("tail_row_cost", C.c_int32), ("sort_queries", C.c_int32), ("host_threads", C.c_int32), ("host_pipeline", C.c_int32),
                ("host_keys", C.c_int32), ("host_pipe_chunk_log2", C.c_int32), ("host_d2h_staged", C.c_int32),
                ("rank_units", C.c_int32), ("marks_32bit", C.c_int32), ("context_mid_table", C.c_int32), ("wavelet_lines", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        lib().femto_amd_options_init(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise KeyError(k)
            setattr(self, k, int(v))


class NfaStruct(C.Structure):
    """femto_amd_nfa_t (include/femto_amd.h): the reference's nfa_description_t, flat"""
    _fields_ = [("num_nodes", C.c_int32), ("num_transitions", C.c_int32), ("trans_start", C.c_void_p),
                ("trans_char", C.c_void_p), ("trans_dest", C.c_void_p), ("is_start", C.c_void_p), ("is_final", C.c_void_p),
                ("cost_bound", C.c_int32), ("subst_cost", C.c_int32), ("delete_cost", C.c_int32), ("insert_cost", C.c_int32)]


class Nfa:
    """An epsilon-free automaton of the REVERSED pattern, as do_regexp_query simulates it (src/main/nfa.h:62-88): node i's
    transitions are entries trans_start[i] .. trans_start[i+1]-1 of trans_char (alpha codes) / trans_dest; settings =
    (cost_bound, subst_cost, delete_cost, insert_cost), cost_bound 1 = exact."""

    def __init__(self, trans_start, trans_char, trans_dest, is_start, is_final, settings=(1, 1, 1, 1)):
        self.trans_start = np.ascontiguousarray(trans_start, dtype=np.int32)
        self.trans_char = np.ascontiguousarray(trans_char, dtype=np.int32)
        self.trans_dest = np.ascontiguousarray(trans_dest, dtype=np.int32)
        self.is_start = np.ascontiguousarray(is_start, dtype=np.uint8)
        self.is_final = np.ascontiguousarray(is_final, dtype=np.uint8)
        self.settings = tuple(int(v) for v in settings)
        self.num_nodes = len(self.is_start)

    @classmethod
    def from_regex(cls, regex, approx=None):
        """femto_amd_regexp_compile: pattern text -> the position automaton of the reversed pattern; approx = (max_cost,
        subst_cost, delete_cost, insert_cost) as in QUERY_FORMAT.txt's APPROX"""
        rx = np.frombuffer(bytes(regex) + b"\0", dtype=np.uint8)
        k = (0, 1, 1, 1) if approx is None else tuple(int(v) for v in approx)
        h = C.c_void_p()
        _check(lib().femto_amd_regexp_compile(_ptr(rx), len(regex), k[0], k[1], k[2], k[3], C.byref(h)))
        try:
            v = lib().femto_amd_regexp_nfa(h).contents
            n, t = v.num_nodes, v.num_transitions

            def arr(p, count, ct):
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(count,)).copy() if count else np.zeros(0, dtype=ct)

            return cls(arr(v.trans_start, n + 1, C.c_int32), arr(v.trans_char, t, C.c_int32), arr(v.trans_dest, t, C.c_int32),
                       arr(v.is_start, n, C.c_uint8), arr(v.is_final, n, C.c_uint8),
                       (v.cost_bound, v.subst_cost, v.delete_cost, v.insert_cost))
        finally:
            lib().femto_amd_regexp_free(h)

    @classmethod
    def from_query(cls, query, icase=False, streamline=True):
        """femto_amd_query_compile: a whole femto_search query prepared as search_tool.cc prepares it (streamline_query,
        simplify_query, --icase).  Returns (Nfa, literal alpha codes or None, the query echoed as ast_to_string prints it)."""
        q = np.frombuffer(bytes(query) + b"\0", dtype=np.uint8)
        h = C.c_void_p()
        _check(lib().femto_amd_query_compile(_ptr(q), len(query), (1 if icase else 0) | (0 if streamline else 2), C.byref(h)))
        try:
            v = lib().femto_amd_regexp_nfa(h).contents
            n, t = v.num_nodes, v.num_transitions

            def arr(p, count, ct):
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(count,)).copy() if count else np.zeros(0, dtype=ct)

            a = cls(arr(v.trans_start, n + 1, C.c_int32), arr(v.trans_char, t, C.c_int32), arr(v.trans_dest, t, C.c_int32),
                    arr(v.is_start, n, C.c_uint8), arr(v.is_final, n, C.c_uint8),
                    (v.cost_bound, v.subst_cost, v.delete_cost, v.insert_cost))
            sp, sn = C.c_void_p(), C.c_int64(0)
            lit = None
            if lib().femto_amd_regexp_literal(h, C.byref(sp), C.byref(sn)):
                lit = arr(sp, sn.value, C.c_uint16)
            return a, lit, lib().femto_amd_regexp_echo(h)
        finally:
            lib().femto_amd_regexp_free(h)

    def struct(self):
        return NfaStruct(self.num_nodes, len(self.trans_char), self.trans_start.ctypes.data, self.trans_char.ctypes.data,
                         self.trans_dest.ctypes.data, self.is_start.ctypes.data, self.is_final.ctypes.data, *self.settings)


class NfaBatch:
    """automata marshalled once into the C ABI's array of femto_amd_nfa_t (what a C caller holds anyway): timing
    Index.nfa_search_batch(NfaBatch) times the library call, not Python building 20 000 structs"""

    def __init__(self, nfas):
        self.nfas = list(nfas)          # keeps the numpy arrays the structs point into alive
        self.n = len(self.nfas)
        self.arr = (NfaStruct * max(1, self.n))(*[a.struct() for a in self.nfas])


_lib = None


def lib():
    """Load (never silently substitute) the HIP extension."""
    global _lib
    if _lib is None:
        if not os.path.exists(_build.LIB):
            raise ImportError(f"{_build.LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        # Load order matters in a process that also uses PyTorch-ROCm: torch ships its own libamdhip64, and two HIP runtimes
        # in one process do not see each other's devices.  Importing torch first makes this library resolve to the runtime
        # torch already loaded (observed on the GPU box: loading this library first made hipGetDeviceCount fail later).
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(_build.LIB)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.femto_amd_open.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        L.femto_amd_close.argtypes = [vp]
        L.femto_amd_close.restype = None
        L.femto_amd_last_error.restype = C.c_char_p
        L.femto_amd_info.argtypes = [vp, C.POINTER(Info)]
        L.femto_amd_count_flat.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        L.femto_amd_count_bytes.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        L.femto_amd_locate_flat.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp, i64, C.POINTER(i64)]
        L.femto_amd_locate_flat_alloc.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, C.POINTER(vp), C.POINTER(i64)]
        L.femto_amd_parallel_locate_range.argtypes = [vp, i64, i64, vp]
        L.femto_amd_parallel_count.argtypes = [vp, i32, vp, vp, vp, vp]
        L.femto_amd_parallel_locate.argtypes = [vp, i32, vp, vp, i32, vp, vp]
        L.femto_amd_resolve_location.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(i64)]
        L.femto_amd_document_info.argtypes = [vp, i64, C.POINTER(C.c_char_p), C.POINTER(i64)]
        L.femto_amd_count_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
        L.femto_amd_locate_plan_device.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        L.femto_amd_locate_walk_device.argtypes = [vp, i64, vp, vp, i64, vp, vp]
        L.femto_amd_locate_device.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, vp, vp]
        L.femto_amd_pack_counts_device.argtypes = [vp, i64, vp, vp, vp, vp, i64, vp, vp]
        L.femto_amd_trace_lines.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, C.POINTER(i64)]
        L.femto_amd_open_multi.argtypes = [C.c_char_p, i32, vp, C.POINTER(vp)]
        L.femto_amd_open_multi_striped.argtypes = [C.c_char_p, i32, vp, C.POINTER(vp)]
        L.femto_amd_striped_serve.argtypes = [vp, C.c_char_p, i32]
        L.femto_amd_open_striped_client.argtypes = [C.c_char_p, C.c_char_p, i32, i32, C.POINTER(vp)]
        L.femto_amd_multi_child.argtypes = [vp, i32, C.POINTER(vp)]
        L.femto_amd_device_count.argtypes = [vp]
        L.femto_amd_comm_unique_id.argtypes = [vp]
        L.femto_amd_comm_init.argtypes = [vp, vp, i32, i32]
        L.femto_amd_comm_gather.argtypes = [vp, vp, vp, i64, i32, vp]
        L.femto_amd_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.femto_amd_regexp_search.argtypes = [vp, vp, i64, i64, vp, vp, vp, C.POINTER(i64)]
        L.femto_amd_regexp_search_approx.argtypes = [vp, vp, i64, i32, i32, i32, i32, i64, vp, vp, vp, vp, C.POINTER(i64)]
        L.femto_amd_regexp_match.argtypes = [vp, i64, vp, i64]
        L.femto_amd_options_init.argtypes = [vp]
        L.femto_amd_options_init.restype = None
        L.femto_amd_open_opts.argtypes = [C.c_char_p, i32, vp, C.POINTER(vp)]
        L.femto_amd_host_pipeline_stats.argtypes = [vp, vp]
        L.femto_amd_key_format.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), vp]
        L.femto_amd_pack_keys_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
        L.femto_amd_locate_keys_device.argtypes = [vp, i64, vp, i32, vp, vp, vp, vp, vp, vp, i64, vp, vp]
        L.femto_amd_regexp_compile.argtypes = [vp, i64, i32, i32, i32, i32, C.POINTER(vp)]
        L.femto_amd_regexp_nfa.argtypes = [vp]
        L.femto_amd_regexp_nfa.restype = C.POINTER(NfaStruct)
        L.femto_amd_regexp_free.argtypes = [vp]
        L.femto_amd_regexp_free.restype = None
        L.femto_amd_nfa_search_batch.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
        L.femto_amd_nfa_stats.argtypes = [vp, vp]
        if hasattr(L, "femto_amd_lf_steps_device"):      # (absent from the older builds tools/ab_bench.sh loads beside this one)
            L.femto_amd_lf_steps_device.argtypes = [vp, i64, vp, vp, vp, vp]
        L.femto_amd_key_table_id.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.femto_amd_set_option.argtypes = [vp, C.c_char_p, i32]
        L.femto_amd_block_requests.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        L.femto_amd_kernel_time_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]
        L.femto_amd_kernel_time_reset.argtypes = [vp]
        L.femto_amd_kernel_time_reset.restype = None
        L.femto_amd_kernel_time_enable.argtypes = [vp, i32]
        L.femto_amd_kernel_time_enable.restype = None
        L.femto_amd_build_index.argtypes = [C.c_char_p, i32, vp, vp, vp, C.c_char_p, i32]
        L.femto_amd_build_index_from_sa.argtypes = [C.c_char_p, i32, vp, vp, vp, C.c_char_p, vp]
        L.femto_amd_forward_steps.argtypes = [vp, i64, vp, vp, vp, vp]
        L.femto_amd_resolve_batch.argtypes = [vp, i64, vp, vp, vp]
        L.femto_amd_resolve_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
        L.femto_amd_query_compile.argtypes = [vp, i64, i32, C.POINTER(vp)]
        L.femto_amd_regexp_literal.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.femto_amd_regexp_echo.argtypes = [vp]
        L.femto_amd_regexp_echo.restype = C.c_char_p
        L.femto_amd_query_echo.argtypes = [vp, i64, i32, i32, vp, i64]
        L.femto_amd_set_rank_mode.argtypes = [vp, i32]
        L.femto_amd_get_rank_mode.argtypes = [vp]
        L.femto_amd_pack_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i32)]
        L.femto_amd_structures.argtypes = [vp, vp, i32]
        L.femto_amd_flatten_index.argtypes = [C.c_char_p, C.c_char_p]
        L.femto_amd_bseq_encode.argtypes = [vp, i64, i32, vp, i64, C.POINTER(i64)]
        L.femto_amd_open_split.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
        L.femto_amd_split_export.argtypes = [vp, vp]
        L.femto_amd_split_attach.argtypes = [vp, i32, vp]
        L.femto_amd_split_attach_local.argtypes = [vp, vp]
        L.femto_amd_split_commit.argtypes = [vp]
        L.femto_amd_split_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]
        _lib = L
    return _lib


def _libc_free(p):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(p)


def _check(rc):
    if rc:
        raise FemtoAmdError(rc, lib().femto_amd_last_error().decode(errors="replace"))


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def flatten(patterns):
    """list of uint16 alpha arrays -> (plen int32[n], flat uint16[sum], starts int64[n])"""
    plen = np.array([len(p) for p in patterns], dtype=np.int32)
    starts = np.zeros(len(patterns), dtype=np.int64)
    if len(patterns):
        starts[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    flat = (np.concatenate([np.asarray(p, dtype=np.uint16) for p in patterns])
            if len(patterns) and plen.sum() else np.zeros(1, dtype=np.uint16))
    return plen, np.ascontiguousarray(flat), starts


class Index:
    """A femto index resident in the HBM of one GPU.  device=-1 parses only (host logic tests)."""

    def __init__(self, path, device=0, part=None, nparts=None, devices=None, striped=False, striped_socket=None, timeout_s=600,
                 _borrowed=None, options=None):
        self._h = C.c_void_p()
        self._peers = []   # range-split: the parts attached in-process must outlive this handle's use
        self._owner = None
        if _borrowed is not None:   # replica of a multi-device handle (femto_amd_multi_child): closed with its parent
            self._owner, self._h = _borrowed
        elif striped_socket is not None:   # striped index built by ANOTHER process (femto_amd_open_striped_client)
            _check(lib().femto_amd_open_striped_client(os.fsencode(path), os.fsencode(striped_socket), int(device), int(timeout_s),
                                                       C.byref(self._h)))
        elif devices is not None:     # one handle over several GPUs of this process (femto_amd_open_multi): host-pointer calls only
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            fn = lib().femto_amd_open_multi_striped if striped else lib().femto_amd_open_multi
            _check(fn(os.fsencode(path), len(devices), arr, C.byref(self._h)))
            device = list(devices)
        elif nparts is None and options is not None:    # femto_amd_open_opts: options = Options(...) or a dict of its fields
            o = options if isinstance(options, Options) else Options(**options)
            _check(lib().femto_amd_open_opts(os.fsencode(path), device, C.byref(o), C.byref(self._h)))
        elif nparts is None:
            _check(lib().femto_amd_open(os.fsencode(path), device, C.byref(self._h)))
        else:
            _check(lib().femto_amd_open_split(os.fsencode(path), device, int(part), int(nparts), C.byref(self._h)))
        self.device = device
        self.part, self.nparts = part, nparts
        self.info = Info()
        _check(lib().femto_amd_info(self._h, C.byref(self.info)))

    # ---- range-split index (femto_amd_open_split): part p holds blocks [nb*p/nparts, nb*(p+1)/nparts)
    def split_export(self):
        """128-byte blob (two hipIpcMemHandle_t) other PROCESSES pass to split_attach"""
        blob = C.create_string_buffer(128)
        _check(lib().femto_amd_split_export(self._h, blob))
        return blob.raw

    def split_attach(self, part, blob):
        _check(lib().femto_amd_split_attach(self._h, int(part), C.create_string_buffer(bytes(blob), 128)))

    def split_attach_local(self, owner):
        """attach a part opened in THIS process (same GPU, or another GPU with peer access)"""
        _check(lib().femto_amd_split_attach_local(self._h, owner._h))
        self._peers.append(owner)

    def split_commit(self):
        _check(lib().femto_amd_split_commit(self._h))

    def split_info(self):
        part, nparts, sb, ib = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
        _check(lib().femto_amd_split_info(self._h, C.byref(part), C.byref(nparts), C.byref(sb), C.byref(ib)))
        return {"part": part.value, "nparts": nparts.value, "seg_bytes": sb.value, "image_bytes": ib.value}

    def striped_serve(self, socket_path, nclients):
        """striped handle (devices=[...], striped=True): hand the stripes to `nclients` other processes (blocks until all
        have attached; femto_amd_striped_serve)"""
        _check(lib().femto_amd_striped_serve(self._h, os.fsencode(socket_path), int(nclients)))

    def child(self, i):
        """replica i of a multi-device handle as a single-GPU Index (borrowed: valid until this handle is closed)"""
        h = C.c_void_p()
        _check(lib().femto_amd_multi_child(self._h, int(i), C.byref(h)))
        dev = self.device[i] if isinstance(self.device, (list, tuple)) else self.device
        return Index(None, device=dev, _borrowed=(self, h))

    def close(self):
        if self._owner is not None:
            self._h = C.c_void_p()
            self._owner = None
            return
        if self._h:
            lib().femto_amd_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- host-array API (numpy in, numpy out)
    def count_flat(self, plen, flat, starts):
        n = len(plen)
        plen = np.ascontiguousarray(plen, dtype=np.int32)
        flat = np.ascontiguousarray(flat, dtype=np.uint16)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        first = np.zeros(n, dtype=np.int64)
        last = np.zeros(n, dtype=np.int64)
        _check(lib().femto_amd_count_flat(self._h, n, _ptr(plen), _ptr(flat), _ptr(starts), _ptr(first), _ptr(last)))
        return first, last

    def count(self, patterns):
        return self.count_flat(*flatten(patterns))

    def locate_flat(self, plen, flat, starts, max_occs):
        n = len(plen)
        plen = np.ascontiguousarray(plen, dtype=np.int32)
        flat = np.ascontiguousarray(flat, dtype=np.uint16)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        noccs = np.zeros(n, dtype=np.int32)
        total = C.c_int64(0)
        buf = C.c_void_p()
        _check(lib().femto_amd_locate_flat_alloc(self._h, n, _ptr(plen), _ptr(flat), _ptr(starts), max_occs, _ptr(noccs), None,
                                                 C.byref(buf), C.byref(total)))
        if not total.value:
            return noccs, np.zeros(0, dtype=np.int64)
        try:
            offs = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int64)), shape=(total.value,)).copy()
        finally:
            _libc_free(buf)
        return noccs, offs

    def locate_flat_two_call(self, plen, flat, starts, max_occs):
        """femto_amd_locate_flat's sizing protocol: first call sizes, second fills a caller-provided buffer"""
        n = len(plen)
        plen = np.ascontiguousarray(plen, dtype=np.int32)
        flat = np.ascontiguousarray(flat, dtype=np.uint16)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        noccs = np.zeros(n, dtype=np.int32)
        ostarts = np.zeros(n + 1, dtype=np.int64)
        total = C.c_int64(0)
        _check(lib().femto_amd_locate_flat(self._h, n, _ptr(plen), _ptr(flat), _ptr(starts), max_occs, _ptr(noccs),
                                           _ptr(ostarts), None, 0, C.byref(total)))
        offs = np.zeros(max(1, total.value), dtype=np.int64)
        if total.value:
            _check(lib().femto_amd_locate_flat(self._h, n, _ptr(plen), _ptr(flat), _ptr(starts), max_occs,
                                               _ptr(noccs), _ptr(ostarts), _ptr(offs), total.value, C.byref(total)))
        return noccs, offs[:total.value]

    def locate_range(self, first, last):
        """text offsets of rows first..last (parallel_locate_range)"""
        out = np.zeros(max(0, last - first + 1), dtype=np.int64)
        _check(lib().femto_amd_parallel_locate_range(self._h, int(first), int(last), _ptr(out)))
        return out

    def locate(self, patterns, max_occs):
        return self.locate_flat(*flatten(patterns), max_occs)

    def pack_info(self):
        """small-alphabet packed lines (mode 3): {'available', 'bytes', 'build_ms', 'ktab_syms'}"""
        a, b, ms, k = C.c_int(0), C.c_int64(0), C.c_double(0), C.c_int(0)
        _check(lib().femto_amd_pack_info(self._h, C.byref(a), C.byref(b), C.byref(ms), C.byref(k)))
        return {"available": bool(a.value & 1), "available2": bool(a.value & 2), "level_table": bool(a.value & 4),
                "sa_full": bool(a.value & 8), "isa_full": bool(a.value & 16), "char_rank_lines": bool(a.value & 32), "context_table": bool(a.value & 64),
                "context_syms": (a.value >> 8) & 15, "context2_syms": (a.value >> 12) & 31, "context_mid_syms": (a.value >> 24) & 31, "rank_units": bool(a.value & (1 << 20)), "rank_units_marked": bool(a.value & (1 << 21)), "sa_32bit": bool(a.value & (1 << 22)),
                "bytes": b.value, "build_ms": ms.value, "ktab_syms": k.value}

    def structures(self):
        """femto_amd_structures: bytes of every structure the handle holds in HBM"""
        out = (C.c_int64 * 16)()
        _check(lib().femto_amd_structures(self._h, out, 16))
        names = ["image", "packed_lines", "marks", "rank_units", "level_table", "context_tables", "char_rank_lines", "text_sa_isa",
                 "two_level_lines", "derived_total", "mark_every", "level_table_syms", "mark_offset_bytes", "hbm_allocated", "hbm_budget",
                 "hbm_budget_is_default"]
        return {k: int(out[i]) for i, k in enumerate(names)}

    def document_info(self, doc):
        p, n = C.c_char_p(), C.c_int64(0)
        pv = C.c_void_p()
        _check(lib().femto_amd_document_info(self._h, int(doc), C.cast(C.byref(pv), C.POINTER(C.c_char_p)), C.byref(n)))
        return C.string_at(pv.value, n.value) if n.value else b""

    def block_requests(self, rows, ch_in=None, location=True):
        """block_request CHAR / OCCS / LOCATION per row: (L[row], occs in block, mark offset or -1); location=False asks for CHAR and
        OCCS only (off comes back None) -- in modes 3 / 4 such a call reads the derived lines alone"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        n = len(rows)
        ch = np.zeros(n, dtype=np.uint16)
        occ = np.zeros(n, dtype=np.int32)
        off = np.zeros(n, dtype=np.int64) if location else None
        chin = np.ascontiguousarray(ch_in, dtype=np.uint16) if ch_in is not None else None
        _check(lib().femto_amd_block_requests(self._h, n, _ptr(rows), _ptr(chin), _ptr(ch), _ptr(occ), _ptr(off)))
        return ch, occ, off

    def forward_steps(self, rows):
        """do_forward_query per row: (ch, LF^-1 row or -1, mark offset or -1)"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        n = len(rows)
        ch = np.zeros(n, dtype=np.uint16)
        nr = np.zeros(n, dtype=np.int64)
        off = np.zeros(n, dtype=np.int64)
        _check(lib().femto_amd_forward_steps(self._h, n, _ptr(rows), _ptr(ch), _ptr(nr), _ptr(off)))
        return ch, nr, off

    def resolve_location(self, offset):
        d, o = C.c_int64(), C.c_int64()
        _check(lib().femto_amd_resolve_location(self._h, offset, C.byref(d), C.byref(o)))
        return d.value, o.value

    def resolve_batch(self, offsets):
        """femto_amd_resolve_batch: (document int64[], offset in document int64[]) for host offsets, searched on the GPU"""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        doc, off = np.zeros(len(offsets), dtype=np.int64), np.zeros(len(offsets), dtype=np.int64)
        _check(lib().femto_amd_resolve_batch(self._h, len(offsets), _ptr(offsets), _ptr(doc), _ptr(off)))
        return doc, off

    def resolve_device(self, d_offsets, n, d_doc=0, d_doc32=0, d_doc_offset=0, d_n=0, stream=0):
        """femto_amd_resolve_device (raw device pointers; enqueue-only)"""
        _check(lib().femto_amd_resolve_device(self._h, d_offsets, n, d_n or None, d_doc or None, d_doc32 or None, d_doc_offset or None,
                                              stream or None))

    # ---- device-pointer API (raw pointers, e.g. torch tensors' data_ptr())
    def count_device(self, npats, d_plen, d_pats, d_starts, d_first, d_last, stream=0):
        _check(lib().femto_amd_count_device(self._h, npats, d_plen, d_pats, d_starts, d_first, d_last or None,
                                            stream or None))

    def locate_plan_device(self, npats, d_plen, d_pats, d_starts, max_occs, d_first, d_last, d_noccs, d_out_starts,
                           stream=0):
        _check(lib().femto_amd_locate_plan_device(self._h, npats, d_plen, d_pats, d_starts, max_occs, d_first, d_last,
                                                  d_noccs, d_out_starts, stream or None))

    def pack_counts_device(self, npats, d_first, d_last, d_counts8, d_big, big_capacity, d_big_n, stream=0):
        """match counts as one byte per pattern + (pattern, count) pairs for counts >= 255 (femto_amd_pack_counts_device)"""
        _check(lib().femto_amd_pack_counts_device(self._h, npats, d_first, d_last, d_counts8, d_big or None, big_capacity, d_big_n,
                                                  stream or None))

    def locate_walk_device(self, npats, d_first, d_out_starts, total, d_offsets, stream=0):
        _check(lib().femto_amd_locate_walk_device(self._h, npats, d_first, d_out_starts, total, d_offsets,
                                                  stream or None))

    def locate_device(self, npats, d_plen, d_pats, d_starts, max_occs, d_first, d_last, d_noccs, d_out_starts, d_offsets,
                      capacity, d_total, stream=0):
        """count + clamp + prefix sum + locate walk in ONE enqueue-only call (femto_amd_locate_device)"""
        _check(lib().femto_amd_locate_device(self._h, npats, d_plen, d_pats, d_starts, max_occs, d_first, d_last, d_noccs,
                                             d_out_starts, d_offsets, capacity, d_total, stream or None))

    TRACE_REGIONS = ("pack_lines", "level_table", "suffix_array", "level1_lines", "level2_lines", "text", "isa", "rank_units", "char_rank_lines",
                     "context_table")

    def trace_lines(self, npats, d_plen, d_pats, d_starts, max_occs):
        """distinct 128-byte lines per derived array loaded by the count phase and by the locate phase of this batch"""
        cl = np.zeros(10, dtype=np.int64)
        ll = np.zeros(10, dtype=np.int64)
        rows = C.c_int64(0)
        _check(lib().femto_amd_trace_lines(self._h, npats, d_plen, d_pats, d_starts, max_occs, _ptr(cl), _ptr(ll), C.byref(rows)))
        return dict(zip(self.TRACE_REGIONS, cl.tolist())), dict(zip(self.TRACE_REGIONS, ll.tolist())), rows.value

    def trace_reads(self):
        """line READS (not only distinct lines) per derived array of the count / the locate phase of the last trace_lines call"""
        cr = np.zeros(10, dtype=np.int64)
        lr = np.zeros(10, dtype=np.int64)
        _check(lib().femto_amd_trace_reads(self._h, _ptr(cr), _ptr(lr)))
        return dict(zip(self.TRACE_REGIONS, cr.tolist())), dict(zip(self.TRACE_REGIONS, lr.tolist()))

    # ---- multi-process gather of device-resident results (RCCL send/recv, femto_amd_comm_*)
    @staticmethod
    def comm_unique_id():
        blob = C.create_string_buffer(128)
        _check(lib().femto_amd_comm_unique_id(blob))
        return blob.raw

    def comm_init(self, id128, nranks, rank):
        _check(lib().femto_amd_comm_init(self._h, C.create_string_buffer(bytes(id128), 128), int(nranks), int(rank)))

    def comm_info(self):
        """{'nranks', 'rank'} as the RCCL communicator reports them"""
        n, r = C.c_int(0), C.c_int(0)
        _check(lib().femto_amd_comm_info(self._h, C.byref(n), C.byref(r)))
        return {"nranks": n.value, "rank": r.value}

    def comm_gather(self, d_send, d_recv, bytes_per_rank, root=0, stream=0):
        _check(lib().femto_amd_comm_gather(self._h, d_send, d_recv or None, int(bytes_per_rank), int(root), stream or None))

    def key_format(self):
        """(bits per field, symbols per key, field_of_alpha uint8[261]) of femto_amd_key_format"""
        bits, syms = C.c_int32(), C.c_int32()
        table = np.zeros(261, dtype=np.uint8)
        _check(lib().femto_amd_key_format(self._h, C.byref(bits), C.byref(syms), _ptr(table)))
        return bits.value, syms.value, table

    def lf_steps_device(self, n, d_rows, d_next, d_off, stream=0):
        """femto_amd_lf_steps_device on raw device addresses: one step of the locate walk per row (offset if marked, else LF(row))"""
        _check(lib().femto_amd_lf_steps_device(self._h, int(n), d_rows, d_next, d_off, stream or None))

    def key_table_id(self):
        """identity of the key fields (femto_amd_key_table_id): keys built for one id are meaningless to a handle reporting another"""
        v = C.c_uint64(0)
        _check(lib().femto_amd_key_table_id(self._h, C.byref(v)))
        return v.value

    def pack_keys_device(self, npats, d_plen, d_pats, d_starts, d_keys, d_bad, stream=0):
        _check(lib().femto_amd_pack_keys_device(self._h, int(npats), d_plen, d_pats, d_starts, d_keys, d_bad, stream or None))

    def locate_keys_device(self, npats, d_keys, max_occs, d_ranges32, d_first, d_last, d_noccs, d_out_starts, d_offsets, capacity, d_total,
                           stream=0):
        """femto_amd_locate_keys_device on raw device addresses (0 / None = NULL)"""
        _check(lib().femto_amd_locate_keys_device(self._h, int(npats), d_keys, int(max_occs), d_ranges32 or None, d_first or None,
                                                  d_last or None, d_noccs or None, d_out_starts or None, d_offsets or None,
                                                  int(capacity), d_total or None, stream or None))

    def nfa_search_batch(self, nfas, max_results=1 << 20):
        """femto_amd_nfa_search_batch: do_regexp_query (src/main/server.c:1656) for a batch of automata on the GPU.
        Returns (result_start int64[n+1], first, last, match_len, cost, status int32[n]): automaton q's results are entries
        result_start[q] .. result_start[q+1]-1, in the order of the reference's sorted result list."""
        if isinstance(nfas, NfaBatch):
            n, arr = nfas.n, nfas.arr
        else:
            nfas = list(nfas)
            n = len(nfas)
            arr = (NfaStruct * max(1, n))(*[a.struct() for a in nfas])
        start = np.zeros(n + 1, dtype=np.int64)
        status = np.zeros(max(1, n), dtype=np.int32)
        m = max(0, int(max_results))      # 0: count only (result_start and the total; the result arrays come back empty)
        first, last = np.zeros(max(1, m), dtype=np.int64), np.zeros(max(1, m), dtype=np.int64)
        mlen, cost = np.zeros(max(1, m), dtype=np.int32), np.zeros(max(1, m), dtype=np.int32)
        tot = C.c_int64(0)
        self.last_total = 0
        try:
            _check(lib().femto_amd_nfa_search_batch(self._h, n, C.cast(arr, C.c_void_p), m, _ptr(start), _ptr(first), _ptr(last), _ptr(mlen),
                                                    _ptr(cost), _ptr(status), C.byref(tot)))
        finally:
            self.last_total = tot.value      # after ERR_FULL: the max_results to call again with
        k = tot.value if m else 0
        return start, first[:k], last[:k], mlen[:k], cost[:k], status[:n]

    def regexp_search_batch(self, regexes, max_results=1 << 20, approx=None):
        """compile every pattern (Nfa.from_regex) and search them as one batch"""
        return self.nfa_search_batch([Nfa.from_regex(r, approx) for r in regexes], max_results)

    def regexp_search(self, regex, max_results=1 << 20, approx=None):
        """the sorted result list of do_regexp_query for one byte regular expression -- or, approx=(max_cost, subst, delete,
        insert), its APPROX form: (first int64[], last int64[], match_len int32[][, cost int32[]]); a range whose string
        matches is a result and is not extended (the reference's rule), ranges inside another result are dropped"""
        rx = np.frombuffer(bytes(regex) + b"\0", dtype=np.uint8)
        k = (0, 1, 1, 1) if approx is None else tuple(int(v) for v in approx)
        n = C.c_int64(0)
        m = max(1, int(max_results))
        first, last = np.zeros(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
        mlen, cost = np.zeros(m, dtype=np.int32), np.zeros(m, dtype=np.int32)
        _check(lib().femto_amd_regexp_search_approx(self._h, _ptr(rx), len(regex), k[0], k[1], k[2], k[3], m, _ptr(first), _ptr(last),
                                                    _ptr(mlen), _ptr(cost), C.byref(n)))
        if approx is None:
            return first[:n.value], last[:n.value], mlen[:n.value]
        return first[:n.value], last[:n.value], mlen[:n.value], cost[:n.value]

    def nfa_stats(self, thread=False):
        """the last automaton batch (thread=True: of the calling thread): pops, span, workgroup occupancy (femto_amd_nfa_stats)"""
        out = (C.c_double * 8)()
        _check(lib().femto_amd_nfa_stats(None if thread else self._h, out))
        keys = ("automata", "workgroups", "pops", "pops_longest", "busy_s", "span_s", "occupancy", "longest_waited_s")
        return dict(zip(keys, [float(v) for v in out]))

    def set_option(self, name, value):
        """'sort' (suffix-order batches of the wavelet-path kernels), 'regexp_max_iterations', 'regexp_stack_cap'"""
        _check(lib().femto_amd_set_option(self._h, name.encode(), int(value)))

    def set_rank_mode(self, mode):
        """3 / 4 = packed / two-level lines (the defaults where they exist), 1 = lane per query on femto's wavelet tree,
        0 = wavefront-per-query walk of femto's raw tables"""
        _check(lib().femto_amd_set_rank_mode(self._h, mode))

    @property
    def rank_mode(self):
        return lib().femto_amd_get_rank_mode(self._h)

    def kernel_time(self, name):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().femto_amd_kernel_time_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_enable(self, on=True):
        lib().femto_amd_kernel_time_enable(self._h, 1 if on else 0)

    def kernel_time_reset(self):
        lib().femto_amd_kernel_time_reset(self._h)


def regexp_match(regex, s):
    """test hook: the automaton built from `regex` accepts exactly the byte string s (None: syntax error)"""
    rx = np.frombuffer(bytes(regex) + b"\0", dtype=np.uint8)
    sb = np.frombuffer(bytes(s) + b"\0", dtype=np.uint8)
    r = lib().femto_amd_regexp_match(_ptr(rx), len(regex), _ptr(sb), len(s))
    return None if r < 0 else bool(r)


def query_echo(query, streamline=True, usequotes=False, simplify=False, icase=False, dump=False):
    """test hook: the query parsed, then streamline_query / simplify_query / icase_ast as asked (femto_search's order), printed
    back as ast_to_string prints it (None: syntax error); dump=True: the PARSED tree in the form oracle/ref_tool.c `ast` reads"""
    q = np.frombuffer(bytes(query) + b"\0", dtype=np.uint8)
    buf = C.create_string_buffer(64 * len(query) + 4096)
    flags = (1 if streamline else 0) | (2 if simplify else 0) | (4 if icase else 0)
    r = lib().femto_amd_query_echo(_ptr(q), len(query), flags, 2 if dump else int(usequotes), buf, len(buf))
    return None if r < 0 else buf.raw[:r]


def bseq_encode(raw_bytes, bitlen, force_type=0):
    """bseq_construct_forcetype-compatible encoding of a bit string (MSB-first bytes)."""
    raw = np.frombuffer(bytes(raw_bytes) + b"\0", dtype=np.uint8)
    n = C.c_int64(0)
    _check(lib().femto_amd_bseq_encode(_ptr(raw), bitlen, force_type, None, 0, C.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    _check(lib().femto_amd_bseq_encode(_ptr(raw), bitlen, force_type, _ptr(out), n.value, C.byref(n)))
    return out


def _doc_args(docs, infos):
    docs = [np.ascontiguousarray(np.frombuffer(bytes(d), dtype=np.uint8) if not isinstance(d, np.ndarray) else d,
                                 dtype=np.uint8) for d in docs]
    n = len(docs)
    ptrs = (C.c_void_p * n)(*[d.ctypes.data if len(d) else None for d in docs])
    lens = np.array([len(d) for d in docs], dtype=np.int64)
    infos = infos or [""] * n
    iarr = (C.c_char_p * n)(*[i.encode() if isinstance(i, str) else bytes(i) for i in infos])
    return docs, n, ptrs, lens, iarr


def build_index(out_dir, docs, params=None, infos=None, device=0):
    """GPU suffix sort + femto block-file writer (femto_amd_build_index)."""
    keep, n, ptrs, lens, iarr = _doc_args(docs, infos)
    _check(lib().femto_amd_build_index(os.fsencode(out_dir), n, ptrs, _ptr(lens), iarr,
                                       params.encode() if params else None, device))


def build_index_from_sa(out_dir, docs, sa, params=None, infos=None):
    """femto block-file writer from a caller-supplied suffix array of the prepared text."""
    keep, n, ptrs, lens, iarr = _doc_args(docs, infos)
    sa = np.ascontiguousarray(sa, dtype=np.int64)
    _check(lib().femto_amd_build_index_from_sa(os.fsencode(out_dir), n, ptrs, _ptr(lens), iarr,
                                               params.encode() if params else None, _ptr(sa)))


def flatten_index(index_dir, out_path):
    """flatten_index (src/main/index.c:2260): directory index -> single flattened file."""
    _check(lib().femto_amd_flatten_index(os.fsencode(index_dir), os.fsencode(out_path)))

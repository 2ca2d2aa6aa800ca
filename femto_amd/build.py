"""Build the in-tree HIP extension (femto_amd/libfemto_amd.so) with hipcc for gfx950.

Every translation unit is compiled to an object under femto_amd/_obj/ (in parallel, only when it or a header changed) and
the objects are linked into the shared library."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.environ.get("FEMTO_AMD_LIB") or os.path.join(HERE, "libfemto_amd.so")
SOURCES = ["femto_amd_api.hip", "api_host.hip", "api_open.hip", "api_multi.hip", "regexp_search.hip", "resolve.hip", "trace_kernels.hip", "host_index.cpp", "index_builder.cpp", "suffix_sort.hip",
           "query_sort.hip", "host_pack.cpp"]
# host-only translation units compiled as plain C++ (x86 intrinsics; no device pass)
PLAIN_CXX = {"host_pack.cpp"}
CXX_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-pthread"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-result", "-Wno-cuda-compat"]
# experiments only (tools/ab_bench.sh): extra compiler flags and a separate object directory for a second build of the library
FLAGS += os.environ.get("FEMTO_AMD_EXTRA_FLAGS", "").split()
if os.environ.get("FEMTO_AMD_OBJ"):
    OBJ = os.environ["FEMTO_AMD_OBJ"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "femto_amd.h"))
    return hs


def _newest_header():
    return max(os.path.getmtime(h) for h in _headers())


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(src, hdr_time):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return os.path.getmtime(os.path.join(CSRC, src)) > t or hdr_time > t


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if _newest_header() > t:
        return True
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest_header()
    todo = [s for s in SOURCES if force or _stale(s, hdr_time)]

    def compile_one(src):
        if src in PLAIN_CXX:
            cmd = ["hipcc"] + CXX_FLAGS + ["-x", "c++", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        else:
            cmd = ["hipcc"] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "femto_amd_multiquery.cpp")
TOOL = os.path.join(HERE, "femto_amd_multiquery")


SEARCH_SRC = os.path.join(os.path.dirname(HERE), "tools", "femto_amd_search.cpp")
SEARCH = os.path.join(HERE, "femto_amd_search")


def build_tools(force=False, verbose=False):
    """The C++ host programs over the C ABI (femto_multiquery's and femto_search's counterparts); plain g++,
    linked against the library."""
    for src, exe in ((TOOL_SRC, TOOL), (SEARCH_SRC, SEARCH)):
        if not force and os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
            continue
        cmd = ["g++", "-std=c++17", "-O2", "-o", exe, src, "-L" + HERE, "-lfemto_amd", "-Wl,-rpath,$ORIGIN",
               "-Wl,-rpath-link," + HERE, "-Wl,--allow-shlib-undefined"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return TOOL


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_tools(force=True, verbose=True)

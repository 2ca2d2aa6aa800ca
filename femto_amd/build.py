"""Build the in-tree HIP extension (femto_amd/libfemto_amd.so) with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("FEMTO_AMD_LIB") or os.path.join(HERE, "libfemto_amd.so")
SOURCES = ["femto_amd_api.hip", "trace_kernels.hip", "host_index.cpp", "index_builder.cpp", "suffix_sort.hip", "query_sort.hip"]
HEADERS = ["regexp_nfa.hpp", "device_tables.h", "host_index.hpp", "host_pipeline.hpp", "kernels.hip.hpp", "pack_kernels.hip.hpp", "pack2_kernels.hip.hpp", "text_kernels.hip.hpp", "direct_kernels.hip.hpp", "ind_kernels.hip.hpp", "trace_api.hpp", "index_builder.hpp",
           os.path.join("..", "..", "include", "femto_amd.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-Wno-unused-result", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
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

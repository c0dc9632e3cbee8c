"""Builds liblcp2.so (HIP kernels + C ABI) for gfx950 with hipcc, in tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container;
the resulting .so travels to the GPU box with the repository snapshot.

Every translation unit is compiled to an object of its own (in parallel, only when it or a header changed), the
objects are linked, and the device code of the result goes through tools/check_hazards.py: a hand-written carry
chain with too few wait states fails the build instead of slipping past the parity tests.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB = os.path.join(PKG_DIR, "liblcp2.so")
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inc"))]
    deps.append(os.path.join(PKG_DIR, "..", "include", "lcp2.h"))
    return deps


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension (there is no CPU fallback)")
    return hipcc


def check_hazards(lib=LIB):
    tools = os.path.join(PKG_DIR, "..", "tools")
    sys.path.insert(0, tools)
    try:
        import check_hazards as ch
    finally:
        sys.path.remove(tools)
    violations, stats = ch.check_library(lib)
    if violations:
        lines = ["  %s: `%s` -> `%s` at %d wait state(s)" % v for v in violations[:10]]
        raise RuntimeError("hazard check failed for %s (%d violations):\n%s" % (lib, len(violations), "\n".join(lines)))
    return stats


def build_native(force=False, verbose=False):
    """Compile every HIP source into eth-lc-plonky2_amd/liblcp2.so. Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(6, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in sources()]
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs)
    stats = check_hazards(LIB + ".tmp")
    if verbose:
        print("hazard check ok: %d VALU-writes-SGPR/VCC -> VALU-reads pairs, closest %s wait states" % (stats["pairs"], stats["min_wait_states"]))
    os.replace(LIB + ".tmp", LIB)
    return LIB


HOST_DIR = os.path.join(PKG_DIR, "host")
HOST_LIB = os.path.join(PKG_DIR, "liblcp2_host.so")


def build_host(force=False, verbose=False):
    """eth-lc-plonky2_amd/liblcp2_host.so: the C++ host layer (CircuitBuilder, the reference's gadgets, light-client update ingestion,
    witness generation) behind the C entry points of host/lc_capi.h, linked against liblcp2.so.  g++ only (no device code)."""
    build_native(verbose=verbose)
    srcs = sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith((".hpp", ".h"))] + [os.path.join(PKG_DIR, "..", "include", "lcp2.h")]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", HOST_LIB + ".tmp"] + srcs + ["-L", PKG_DIR, "-llcp2", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(HOST_LIB + ".tmp", HOST_LIB)
    return HOST_LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))

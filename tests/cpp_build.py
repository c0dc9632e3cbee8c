"""Builds tests/cpp/test_gadgets: the C++ circuit-level tests (host layer + C ABI + oracle as the CPU checker)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
HOST = os.path.join(ROOT, "eth-lc-plonky2_amd", "host")
BIN = os.path.join(CPP, "test_gadgets")
HOST_SOURCES = ("gates.cpp", "builder.cpp", "gadgets.cpp", "light_client_update.cpp", "poseidon_host.cpp", "recursion.cpp", "biguint.cpp")
EXAMPLE_SRC = os.path.join(ROOT, "examples", "lc_prover.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "lc_prover")


def build():
    import eth_lc_plonky2_amd as m
    import oracle_lib
    m.build_native()
    oracle_lib.build()
    hdr = os.path.join(CPP, "golden_data.hpp")
    gen = os.path.join(CPP, "make_golden_header.py")
    kat = os.path.join(ROOT, "tests", "golden", "sha256_kat.json")
    lcu = os.path.join(ROOT, "tests", "golden", "lc_updates.json")
    if not os.path.exists(hdr) or os.path.getmtime(hdr) < max(os.path.getmtime(gen), os.path.getmtime(kat), os.path.getmtime(lcu)):
        subprocess.run(["python3", gen, hdr], check=True)
    srcs = [os.path.join(CPP, "test_gadgets.cpp")] + [os.path.join(HOST, f) for f in HOST_SOURCES]
    deps = srcs + [hdr] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [
        os.path.join(ROOT, "include", "lcp2.h"), os.path.join(ROOT, "oracle", "plonk.h")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        pkg, orc = os.path.join(ROOT, "eth-lc-plonky2_amd"), os.path.join(ROOT, "oracle")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", BIN] + srcs + ["-L", pkg, "-llcp2", "-L", orc, "-loracle",
                        "-Wl,-rpath," + pkg, "-Wl,-rpath," + orc, "-fopenmp"], check=True)
    return BIN


def run(mode, test, timeout=900):
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "16")
    return subprocess.run([build(), mode, test], capture_output=True, text=True, timeout=timeout, env=env)


def build_example():
    """examples/lc_prover: product code only (host layer + liblcp2.so), no oracle linked."""
    import eth_lc_plonky2_amd as m
    m.build_native()
    srcs = [EXAMPLE_SRC] + [os.path.join(HOST, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "lcp2.h")]
    if not os.path.exists(EXAMPLE_BIN) or any(os.path.getmtime(d) > os.path.getmtime(EXAMPLE_BIN) for d in deps):
        pkg = os.path.join(ROOT, "eth-lc-plonky2_amd")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", EXAMPLE_BIN] + srcs + ["-L", pkg, "-llcp2", "-Wl,-rpath," + pkg], check=True)
    return EXAMPLE_BIN


def run_example(args, timeout=900):
    return subprocess.run([build_example()] + list(args), capture_output=True, text=True, timeout=timeout)

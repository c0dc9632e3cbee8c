"""csrc/generated_gates.hpp is checked in next to its generator (tools/gen/gen_native_gates.cpp, which walks the gate programs of
host/gates.cpp).  A stale file cannot mis-prove - lcp2_circuit_create checks every native claim against the program on random
points and refuses the build() - but it would only be noticed on a GPU.  This regenerates the file on the CPU and diffs it."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_gates_header_is_current():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "gen")
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "gen", "gen_native_gates.cpp"),
                        os.path.join(ROOT, "eth-lc-plonky2_amd", "host", "gates.cpp"), os.path.join(ROOT, "eth-lc-plonky2_amd", "host", "poseidon_host.cpp")],
                       check=True, cwd=ROOT)
        fresh = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    have = open(os.path.join(ROOT, "eth-lc-plonky2_amd", "csrc", "generated_gates.hpp")).read()
    assert fresh == have, "csrc/generated_gates.hpp is stale: run tools/gen/run.sh"
    assert fresh.count("template <> __device__ __forceinline__ void q_generated<") == 4

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


def test_generated_schedules_fit_their_register_budget():
    """the windowed forms are scheduled by the generator for register pressure; the kernels' occupancy bounds (kernels_prover.hip
    k_q_gate: 4 waves per SIMD for the ShaAddGate, 3 for the round gates) assume these peaks: a change of a gate program that raises
    them shows here, not as spills on the GPU.  Every value alive across a window boundary must be pinned there (Q_PIN), and a
    window must not request more than 8 new wires."""
    import re
    text = open(os.path.join(ROOT, "eth-lc-plonky2_amd", "csrc", "generated_gates.hpp")).read()
    peaks = {name: int(v) for name, v in re.findall(r"// (\w+): \d+ instructions, \d+ constraints, \d+ wires and gate constants; at most (\d+) 64-bit values live", text)}
    assert peaks == {"ShaAddGate": 9, "ShaRoundAGate": 54, "ShaRoundEGate": 48}, peaks
    assert "ShaScheduleGate" in text and "(plain form)" in text
    for body in text.split("template <> __device__")[1:4]:
        windows = body.split("Q_WINDOW_BARRIER();")
        assert len(windows) > 8
        for w in windows[1:-1]:
            assert len(re.findall(r"^  u64 [wk]\d+ = ", w, re.M)) <= 8
        assert body.count("terms.pin();") == len(windows) - 1

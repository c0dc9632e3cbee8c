"""csrc/generated_gates*.hpp are checked in next to their generator (tools/gen/gen_native_gates.cpp, which walks the gate programs of
host/gates.cpp and of tools/gen/reference_gate_programs.txt, the dump of the Python gate libraries).  A stale file cannot mis-prove -
lcp2_circuit_create checks every native claim against the program on random points and refuses the build() - but it would only be
noticed on a GPU.  These tests regenerate dump and headers on the CPU and diff them, hold the Python table of generated indices to the
index header, and run every generated evaluator of the dump on the CPU (tests/emu/emu_gates.cpp) against a Python interpretation of
its program."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eth-lc-plonky2_amd", "csrc")
DUMP = os.path.join(ROOT, "tools", "gen", "reference_gate_programs.txt")
UNITS = ("sha", "u32a", "u32b", "reca", "recb")
P = 0xFFFFFFFF00000001


def test_program_dump_is_current():
    fresh = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "dump_reference_programs.py"), "--stdout"], check=True, capture_output=True, text=True).stdout
    assert fresh == open(DUMP).read(), "tools/gen/reference_gate_programs.txt is stale: run tools/gen/run.sh"


def test_generated_gates_headers_are_current():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "gen")
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "gen", "gen_native_gates.cpp"),
                        os.path.join(ROOT, "eth-lc-plonky2_amd", "host", "gates.cpp"), os.path.join(ROOT, "eth-lc-plonky2_amd", "host", "poseidon_host.cpp")],
                       check=True, cwd=ROOT)
        subprocess.run([exe, DUMP, d], check=True)
        names = ["generated_gates.hpp"] + ["generated_gates_%s.hpp" % u for u in UNITS]
        count = 0
        for name in names:
            fresh, have = open(os.path.join(d, name)).read(), open(os.path.join(CSRC, name)).read()
            assert fresh == have, "csrc/%s is stale: run tools/gen/run.sh" % name
            count += fresh.count("template <> __device__ __forceinline__ void q_generated<")
    assert count == 17
    assert sorted(f for f in os.listdir(CSRC) if f.startswith("generated_gates")) == sorted(names)


def test_python_index_matches_the_generated_index():
    """circuit.py GENERATED_GATE_NAMES (what a GateSet claims) against the table of csrc/generated_gates.hpp"""
    import eth_lc_plonky2_amd as m
    text = open(os.path.join(CSRC, "generated_gates.hpp")).read()
    rows = re.findall(r"^// +(\d+) (\w+) +(\w+) ", text, re.M)
    assert [name for _, name, _ in rows] == list(m.circuit.GENERATED_GATE_NAMES)
    assert [int(k) for k, _, _ in rows] == list(range(len(rows)))
    assert int(re.search(r"Q_GENERATED_COUNT = (\d+);", text).group(1)) == len(rows)
    assert "QUOTIENT_GENERATED_GATES = %d;" % len(rows) in open(os.path.join(CSRC, "prover_kernels.hpp")).read()
    for unit in UNITS:  # every unit has its compile unit
        assert os.path.exists(os.path.join(CSRC, "kernels_gates_%s.hip" % unit))


def test_generated_schedules_fit_their_register_budget():
    """the windowed forms are scheduled by the generator for register pressure; the kernels' occupancy bounds (Q_GENERATED_WAVES in the
    index, used by k_q_gen's launch bounds) assume these peaks: a change of a gate program that raises them shows here, not as spills
    on the GPU.  Every value alive across a window boundary must be pinned there (Q_PIN), and a window must not request more than 8
    new wires."""
    text = "".join(open(os.path.join(CSRC, "generated_gates_%s.hpp" % u)).read() for u in UNITS)
    peaks = {name: int(v) for name, v in re.findall(r"// (\w+): \d+ instructions, \d+ constraints, \d+ wires and gate constants; at most (\d+) 64-bit values live", text)}
    assert {k: peaks[k] for k in ("ShaAddGate", "ShaRoundAGate", "ShaRoundEGate")} == {"ShaAddGate": 9, "ShaRoundAGate": 54, "ShaRoundEGate": 48}, peaks
    assert "ShaScheduleGate" in text and "(plain form)" in text
    index = open(os.path.join(CSRC, "generated_gates.hpp")).read()
    waves = [int(v) for v in re.search(r"Q_GENERATED_WAVES\[Q_GENERATED_COUNT\] = \{([^}]*)\}", index).group(1).split(",")]
    assert waves[:4] == [4, 3, 3, 2]
    import eth_lc_plonky2_amd as m
    for k, name in enumerate(m.circuit.GENERATED_GATE_NAMES):
        if name in peaks and k >= 4:  # 2 VGPRs per live value + column sums, loads in flight and temporaries within 512 / waves
            assert peaks[name] <= {4: 24, 3: 50, 2: 100}[waves[k]], (name, peaks[name], waves[k])
    bodies = text.split("template <> __device__")[1:]
    assert len(bodies) == 17
    for body in bodies:
        if "(plain form)" in body.split("\n")[0] or "emit.begin_terms();" in body:
            continue
        windows = body.split("Q_WINDOW_BARRIER();")
        for w in windows[1:-1]:
            assert len(re.findall(r"^  u64 [wk]\d+ = ", w, re.M)) <= 8
        assert body.count("terms.pin();") == len(windows) - 1


# ---------------------------------------------------------------- generated evaluators on the CPU against the programs
def _interpret(code, imm, wires, consts, pis, forward, alpha):
    """include/lcp2.h semantics over Python integers -> sum_j alpha^j constraint_j"""
    reg = [0] * 256
    emitted = []

    def operand(kind, idx):  # 0 REG, 1 WIRE, 2 CONST (gate constant, after the selector columns), 3 IMM, 4 PI
        return (reg, wires, consts, imm, pis)[kind][idx]

    MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    for pc in range(len(code) // 2):
        w0, w1 = int(code[2 * pc]), int(code[2 * pc + 1])
        op, dst, ka, kb, ia, ib = w0 & 0xF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xF, (w0 >> 20) & 0xF, w1 & 0xFFFF, w1 >> 16
        if op == 9:
            src = [reg[ia + j] for j in range(12)]
            for r in range(12):
                reg[dst + r] = (sum(src[(i + r) % 12] * MDS_CIRC[i] for i in range(12)) + (8 * src[0] if r == 0 else 0) + imm[ib + r]) % P
            continue
        x = operand(ka, ia)
        if op == 3:
            emitted.append(x)
            continue
        if op == 6:
            emitted.append((x * x - x) % P)
            continue
        if op == 8:
            reg[dst] = pow(x, 7, P)
            continue
        y = operand(kb, ib)
        reg[dst] = {0: x + y, 1: x - y, 2: x * y, 4: x + y - 2 * x * y, 5: 2 * x + y, 7: reg[dst] + x * y}[op] % P
    if not forward:
        emitted.reverse()
    return sum(c * pow(alpha, j, P) for j, c in enumerate(emitted)) % P


def _programs():
    """(generated index, name, code, imm, forward) of every program in the dump"""
    import eth_lc_plonky2_amd as m
    out, imm = [], None
    for line in open(DUMP):
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        if tok[0] == "gateset":
            imm = [int(v, 16) for v in tok[3:3 + int(tok[2], 16)]]
        elif tok[0] == "gate":
            flags, length = int(tok[2], 16), int(tok[4], 16)
            out.append((m.circuit.GENERATED_GATE_INDEX[tok[1]], tok[1], [int(v, 16) for v in tok[5:5 + 2 * length]], imm, bool(flags & 1)))
    return out


def test_generated_evaluators_equal_their_programs_on_the_cpu():
    """every evaluator generated from the dump (lazy forms, shift / 32-bit constant multiplications, rewritten range products, MDS rows,
    explicit alpha exponents) against the interpretation of its program: 6 random points, two challenges, then alpha = 0 (only the
    first constraint survives: the corner where a wrong constraint order shows)"""
    import ctypes
    import emu_lib
    E = emu_lib.load_gates()
    progs = _programs()
    assert E.emu_generated_count() == 17 and sorted(k for k, *_ in progs) == list(range(4, 17))
    rng = np.random.default_rng(2024)
    count, nw, nsel, nk = 6, 135, 3, 2
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for k, name, code, imm, forward in progs:
        wires = rng.integers(0, P, size=(nw, count), dtype=np.uint64)
        wires[:, 0] = rng.integers(0, 4, size=nw).astype(np.uint64)       # a point of small values: digits, where range products vanish
        wires[:, 1] = np.uint64(P - 1)                                      # and the largest canonical value everywhere
        consts = rng.integers(0, P, size=(nsel + nk, count), dtype=np.uint64)
        pis = rng.integers(0, P, size=4, dtype=np.uint64)
        for alphas in ([int(rng.integers(1, P, dtype=np.uint64)), int(rng.integers(1, P, dtype=np.uint64))], [0, 3]):
            al = np.array(alphas, dtype=np.uint64)
            out = np.zeros((count, 2), dtype=np.uint64)
            assert E.emu_generated_gate(k, vp(wires), vp(consts), vp(pis), nsel, count, vp(al), vp(out)) == 0
            for i in range(count):
                w = [int(v) for v in wires[:, i]]
                c = [int(v) for v in consts[nsel:, i]]
                for ch in range(2):
                    want = _interpret(code, imm, w, c, [int(v) for v in pis], forward, alphas[ch])
                    assert int(out[i, ch]) == want, (name, i, ch, alphas)


def test_constant_multiplication_helpers():
    import emu_lib
    E = emu_lib.load_gates()
    rng = np.random.default_rng(7)
    xs = [0, 1, P - 1, P, 2 ** 64 - 1, 2 ** 32, 2 ** 32 - 1] + [int(v) for v in rng.integers(0, 2 ** 64, size=200, dtype=np.uint64)]
    for x in xs:
        for c in (2, 3, 7, 0xFFFFFFFF, 0x80000001, int(rng.integers(2, 2 ** 32))):
            assert E.emu_gl_mul_u32(x, c) == x * c % P
        for s in (1, 2, 31, 32):
            assert E.emu_gl_shl_nc(x, s) == (x << s) % P

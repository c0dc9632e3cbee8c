"""Builds/loads tests/emu/libemu.so: CPU emulation of the device kernels' phases (test harness only)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "emu.cpp")
LIB = os.path.join(HERE, "emu", "libemu.so")
CSRC = os.path.join(HERE, "..", "eth-lc-plonky2_amd", "csrc")


def load():
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC], check=True)
    E = ctypes.CDLL(LIB)
    c = ctypes
    V = c.c_void_p
    E.emu_poseidon_permute.argtypes = [V]
    E.emu_poseidon_permute_grouped.argtypes = [V]
    E.emu_poseidon_partial_max_entry.restype = c.c_uint32
    E.emu_gl_mul.restype = c.c_uint64
    E.emu_gl_mul.argtypes = [c.c_uint64, c.c_uint64]
    E.emu_gl_shl.restype = c.c_uint64
    E.emu_gl_shl.argtypes = [c.c_uint64, c.c_uint]
    E.emu_ntt_forward.restype = c.c_int
    E.emu_ntt_forward.argtypes = [V, V, c.c_uint32, c.c_uint32, c.c_uint64, c.c_uint32]
    E.emu_ntt_inverse_natural.argtypes = [V, V, c.c_uint32, c.c_uint32]
    E.emu_ntt_inverse_bitrev.argtypes = [V, V, c.c_uint32, c.c_uint32, c.c_uint64]
    return E


GATES_SRC = os.path.join(HERE, "emu", "emu_gates.cpp")
GATES_LIB = os.path.join(HERE, "emu", "libemu_gates.so")


def load_gates():
    """tests/emu/libemu_gates.so: the generated gate evaluators (csrc/generated_gates_*.hpp) compiled for the CPU"""
    deps = [GATES_SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("generated_gates") or f in ("gate_helpers.hpp", "gl64.hpp")]
    if not os.path.exists(GATES_LIB) or any(os.path.getmtime(d) > os.path.getmtime(GATES_LIB) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", GATES_LIB, GATES_SRC], check=True)
    E = ctypes.CDLL(GATES_LIB)
    c = ctypes
    V = c.c_void_p
    E.emu_generated_count.restype = c.c_uint
    E.emu_generated_waves.restype = c.c_uint
    E.emu_generated_waves.argtypes = [c.c_uint]
    E.emu_generated_gate.restype = c.c_int
    E.emu_generated_gate.argtypes = [c.c_uint, V, V, V, c.c_uint, c.c_uint64, V, V]
    E.emu_gl_mul_u32.restype = c.c_uint64
    E.emu_gl_mul_u32.argtypes = [c.c_uint64, c.c_uint32]
    E.emu_gl_shl_nc.restype = c.c_uint64
    E.emu_gl_shl_nc.argtypes = [c.c_uint64, c.c_uint]
    return E

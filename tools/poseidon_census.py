#!/usr/bin/env python3
"""Instruction census of one Poseidon permutation as compiled into liblcp2.so, weighted by the issue rates measured with
tools/ubench/int_rates (profiles/r02_ubench_int_rates.txt): the opcode-weighted issue floor of the hash kernels.

The permutation inside k_poseidon_permute_batch is three rolled loops (4 full rounds, 7 groups of three partial rounds, 3 full rounds), one peeled partial round and an
unrolled last round; the loops are found as backward s_cbranch edges and their bodies are multiplied by the trip counts.
    python tools/poseidon_census.py [liblcp2.so] > profiles/r02_poseidon_census.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_hazards as ch  # noqa: E402

TRIPS = [4, 7, 3]  # POS_FULL_HALF, POS_GROUPS (three partial rounds per trip; the 22nd partial round is peeled), POS_FULL_HALF - 1 (the last full round is peeled)
INSN = re.compile(r"^\s+([a-z][a-z0-9_]*)\s*(.*?)\s*//\s*([0-9A-F]+):")


def issue_class(mn):
    """classes of profiles/r02_ubench_int_rates.txt (cycles per wave-instruction per SIMD at 8 waves per SIMD)"""
    if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64"):
        return "mad64"
    if mn in ("v_mov_b32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
              "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_cndmask_b32_e32"):
        return "vop2_plain"
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32", mn):
        return "carry"
    if mn.startswith("v_"):
        return "vop3_other"
    if mn == "s_nop":
        return "s_nop"
    if mn.startswith("s_load") or mn.startswith("s_buffer"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def census(lib):
    text = None
    for t in ch.device_disassembly(lib):
        if "k_poseidon_permute_batch" in t:
            text = t
    m = re.search(r"<_ZN4lcp224k_poseidon_permute_batch[^>]*>:(.*?)s_endpgm", text, re.S)
    ins = []
    for line in m.group(1).splitlines():
        mm = INSN.match(line)
        if mm:
            ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
    addr = [a for a, _, _ in ins]
    weight = [1] * len(ins)
    loops = []
    for k, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch"):
            # llvm-objdump prints the branch target as a word offset; recompute: target = next pc + simm16 * 4
            off = int(ops.split()[0], 0)
            off = off - 0x10000 if off >= 0x8000 else off
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in addr:
                loops.append((addr.index(tgt), k))
    loops.sort()
    assert len(loops) == len(TRIPS), "expected %d rolled loops in the permutation, found %d" % (len(TRIPS), len(loops))
    for (lo, hi), trips in zip(loops, TRIPS):
        for k in range(lo, hi + 1):
            weight[k] *= trips
    counts = {}
    for (a, mn, ops), w in zip(ins, weight):
        c = issue_class(mn)
        counts[c] = counts.get(c, 0) + w
        if mn == "s_nop":
            counts["s_nop_wait_states"] = counts.get("s_nop_wait_states", 0) + w * (int(ops.split()[0], 0) + 1)
    return counts, [(hi - lo + 1) for lo, hi in loops]


def ubench_rates(path):
    r = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 4 and f[0].startswith("k_"):
            r[f[0]] = float(f[3])
    return r


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "eth-lc-plonky2_amd", "liblcp2.so")
    counts, loop_sizes = census(lib)
    ub = ubench_rates(os.path.join(ROOT, "profiles", "r02_ubench_int_rates.txt"))
    # k_cndmask* of the ubench reads a vcc nothing writes and is not a usable figure (22 cycles); v_cndmask_b32_e32 is a VOP2
    # like v_add_u32 and is priced as one
    cyc = {"mad64": ub["k_mad_u64_u32_inl"], "carry": ub["k_add_co_u32"], "vop2_plain": ub["k_add_u32"], "vop3_other": ub["k_add3_u32"]}
    valu = sum(counts.get(c, 0) for c in cyc)
    floor_cycles = sum(counts.get(c, 0) * cyc[c] for c in cyc)
    out = {"source": "llvm-objdump of k_poseidon_permute_batch in liblcp2.so, loop bodies x trip counts %s (body sizes %s instructions)" % (TRIPS, loop_sizes),
           "per_kernel_invocation_of_one_permutation": counts, "valu_instructions": valu,
           "ubench_cycles_per_wave_instruction": cyc, "issue_floor_cycles_per_permutation_per_wave": floor_cycles,
           "average_cycles_per_valu_instruction_at_the_floor": floor_cycles / valu}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

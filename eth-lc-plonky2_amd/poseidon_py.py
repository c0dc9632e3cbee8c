"""Poseidon-12 over Goldilocks on the host, for circuit construction only: the witness of a PoseidonGate row and the
in-circuit public-input hash (plonky2 0.1.4 gates/poseidon.rs `PoseidonGenerator`, hash/hashing.rs
`hash_n_to_hash_no_pad`, plonk/circuit_builder.rs::build).  A handful of rows per circuit: plain Python integers.
The prover's hashing runs in the HIP kernels (csrc/poseidon.hpp); this file is never on the hot path.
"""
P = 0xFFFFFFFF00000001
WIDTH, N_FULL_HALF, N_PARTIAL, N_ROUNDS = 12, 4, 22, 30
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11

# PoseidonGate wire layout (gates/poseidon.rs)
W_INPUT, W_OUTPUT, W_SWAP, W_DELTA = 0, 12, 24, 25
START_FULL_0, START_PARTIAL, START_FULL_1, NUM_WIRES = 29, 65, 87, 135


def wire_full_sbox_0(r, i):
    assert 1 <= r < N_FULL_HALF
    return START_FULL_0 + WIDTH * (r - 1) + i


def wire_partial_sbox(r):
    return START_PARTIAL + r


def wire_full_sbox_1(r, i):
    return START_FULL_1 + WIDTH * r + i


_RC = None


def round_constants():
    """ChaCha8Rng::seed_from_u64(0) sampled with rand-0.8 Uniform::new(0, p): 360 values"""
    global _RC
    if _RC is not None:
        return _RC
    M32 = 0xFFFFFFFF

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & M32

    key, st = [], 0
    for _ in range(8):
        st = (st * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
        xs = (((st >> 18) ^ st) >> 27) & M32
        rot = st >> 59
        key.append(((xs >> rot) | (xs << (32 - rot))) & M32 if rot else xs)

    def block(ctr):
        inp = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + key + [ctr & M32, ctr >> 32, 0, 0]
        x = list(inp)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 7)

        for _ in range(4):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return [(x[i] + inp[i]) & M32 for i in range(16)]

    out, words, ctr = [], [], 0
    while len(out) < N_ROUNDS * WIDTH:
        while len(words) < 2:
            words += block(ctr)
            ctr += 1
        v = words[0] | (words[1] << 32)
        words = words[2:]
        m = v * P
        if (m & 0xFFFFFFFFFFFFFFFF) <= 0xFFFFFFFF00000000:
            out.append(m >> 64)
    _RC = out
    return out


def mds(s):
    return [(sum(s[(i + r) % 12] * MDS_CIRC[i] for i in range(12)) + s[r] * MDS_DIAG[r]) % P for r in range(12)]


def permute_trace(state):
    """The permutation in its naive round form, returning (output, sbox inputs per round).  The value entering lane 0's S-box in
    a partial round is the same in plonky2's fast-partial-round refactoring, which is what its generator writes to the wire."""
    rc = round_constants()
    s = [int(v) % P for v in state]
    sbox_in = []
    for r in range(N_ROUNDS):
        s = [(s[i] + rc[12 * r + i]) % P for i in range(12)]
        full = r < N_FULL_HALF or r >= N_FULL_HALF + N_PARTIAL
        sbox_in.append(list(s) if full else [s[0]])
        if full:
            s = [pow(v, 7, P) for v in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = mds(s)
    return s, sbox_in


def permute(state):
    return permute_trace(state)[0]


def hash_no_pad(values):
    s = [0] * 12
    vals = [int(v) % P for v in values]
    for off in range(0, len(vals), 8):
        chunk = vals[off:off + 8]
        s[:len(chunk)] = chunk
        s = permute(s)
    return s[:4]


def gate_row(inputs, swap=0):
    """The 135 wire values of one PoseidonGate row (PoseidonGenerator::run_once)."""
    inp = [int(v) % P for v in inputs]
    assert len(inp) == 12 and swap in (0, 1)
    w = [0] * NUM_WIRES
    w[W_INPUT:W_INPUT + 12] = inp
    w[W_SWAP] = swap
    state = list(inp)
    for i in range(4):
        delta = swap * (inp[i + 4] - inp[i]) % P
        w[W_DELTA + i] = delta
        state[i] = (inp[i] + delta) % P
        state[i + 4] = (inp[i + 4] - delta) % P
    out, sbox_in = permute_trace(state)
    for r in range(1, N_FULL_HALF):
        for i in range(12):
            w[wire_full_sbox_0(r, i)] = sbox_in[r][i]
    for r in range(N_PARTIAL):
        w[wire_partial_sbox(r)] = sbox_in[N_FULL_HALF + r][0]
    for r in range(N_FULL_HALF):
        for i in range(12):
            w[wire_full_sbox_1(r, i)] = sbox_in[N_FULL_HALF + N_PARTIAL + r][i]
    w[W_OUTPUT:W_OUTPUT + 12] = out
    return w

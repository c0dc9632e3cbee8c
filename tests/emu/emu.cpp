// TEST HARNESS (not product code): runs the portable load/step/store phases of the
// device kernels on the CPU, thread by thread, so that index logic and field
// arithmetic of csrc/*.hpp can be checked against the oracle without a GPU.
// Built by tests/emu_lib.py with g++; never loaded by the package.
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include "../../eth-lc-plonky2_amd/csrc/poseidon.hpp"
#include "../../eth-lc-plonky2_amd/csrc/ntt_host.hpp"

using namespace lcp2;

struct EmuBackend {
  std::map<std::string, std::vector<u64>> tabs;
  const u64 *table(const std::string &k, std::function<std::vector<u64>()> make) {
    auto it = tabs.find(k);
    if (it == tabs.end()) it = tabs.emplace(k, make()).first;
    return it->second.data();
  }
  void launch_pass(bool inv, const NttPassParams &p, u32 wgs, u32 cols, u32 nz) {
    NttPass pass{p};
    std::vector<u64> lds(ntt_lds_words(p.L));
    const u32 T = NTT_THREADS;
    for (u32 z = p.z_base; z < p.z_base + nz; z++)
      for (u32 c = 0; c < cols; c++)
        for (u32 w = 0; w < wgs; w++) {
          for (u32 t = 0; t < T; t++) inv ? pass.load<true>(lds.data(), t, T, w, c, z) : pass.load<false>(lds.data(), t, T, w, c, z);
          for (u32 si = 0; si < p.nsteps; si++)
            for (u32 t = 0; t < T; t++) inv ? pass.step<true>(lds.data(), t, T, si) : pass.step<false>(lds.data(), t, T, si);
          for (u32 t = 0; t < T; t++) inv ? pass.store<true>(lds.data(), t, T, w, c, z) : pass.store<false>(lds.data(), t, T, w, c, z);
        }
  }
  void launch_bitrev(const BitrevTile &b, u32 wgs, u32 cols) {
    std::vector<u64> lds(64 * 65);
    for (u32 c = 0; c < cols; c++)
      for (u32 w = 0; w < wgs; w++) {
        for (u32 t = 0; t < 256; t++) b.load(lds.data(), t, 256, w, c);
        for (u32 t = 0; t < 256; t++) b.store(lds.data(), t, 256, w, c);
      }
  }
  void launch_bitrev_small(const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols) {
    for (u32 c = 0; c < cols; c++)
      for (u32 i = 0; i < (1u << lg); i++) out[c * os + bitrev32(i, lg)] = gl_canon(in[c * is + i]);
  }
};

static u64 g_rc[360];
static bool g_rc_ok = false;

extern "C" {
void emu_poseidon_permute(u64 *s) {
  if (!g_rc_ok) { pos_derive_round_constants(g_rc); g_rc_ok = true; }
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
  pos_permute(s, g_rc);
}
u64 emu_gl_mul(u64 a, u64 b) { return gl_mul(gl_canon(a), gl_canon(b)); }
// x * 2^s for the shifts the NTT's register butterflies use (csrc/gl64.hpp gl_shl); x canonical
u64 emu_gl_shl(u64 x, unsigned s) {
  switch (s) {
    case 12: return gl_shl<12>(x);
    case 24: return gl_shl<24>(x);
    case 32: return gl_shl<32>(x);
    case 33: return gl_shl<33>(x);
    case 36: return gl_shl<36>(x);
    case 48: return gl_shl<48>(x);
    case 60: return gl_shl<60>(x);
    case 63: return gl_shl<63>(x);
    case 65: return gl_shl<65>(x);
    case 72: return gl_shl<72>(x);
    case 84: return gl_shl<84>(x);
    case 95: return gl_shl<95>(x);
    default: return ~0ull;
  }
}
void emu_ntt_forward(const u64 *in, u64 *out, u32 lg, u32 ncols, u64 shift, u32 zbits) {
  EmuBackend be; NttHost<EmuBackend> h(be);
  h.forward(in, (u64)1 << lg, out, (u64)1 << (lg + zbits), lg, ncols, shift, zbits);
}
void emu_ntt_inverse_natural(const u64 *in, u64 *out, u32 lg, u32 ncols) {
  EmuBackend be; NttHost<EmuBackend> h(be);
  h.inverse_natural(in, (u64)1 << lg, out, (u64)1 << lg, lg, ncols);
}
void emu_ntt_inverse_bitrev(const u64 *in, u64 *out, u32 lg, u32 ncols, u64 shift) {
  EmuBackend be; NttHost<EmuBackend> h(be);
  h.inverse_bitrev_in(in, (u64)1 << lg, out, (u64)1 << lg, lg, ncols, shift);
}
}

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
  int pf_launches = 0;  // passes that took the prefetching kernel's phases
  void launch_pass(bool inv, const NttPassParams &p, u32 wgs, u32 cols, u32 nz) {
    NttPass pass{p};
    std::vector<u64> lds(ntt_lds_words(p.L));
    const u32 T = NTT_THREADS;
    if (ntt_pf_slabs_per_wg(p, inv, (u64)wgs * cols * nz)) {
      // k_ntt_pass_pf thread by thread: every phase for all threads before the next one (= the barriers of the kernel; the two
      // halves of the wave-local store of the contiguous pass are separated the same way)
      pf_launches++;
      const bool strided = ntt_pf_strided(p), s7 = ntt_pf_strided_s(p) == 7;
      const int fmode = !strided || !p.scale_mode ? 0 : (p.sc_step ? 2 : 1);
      std::vector<u64> regs((size_t)T * 16), twl(NttPass::PF_TW_WORDS_CONTIGUOUS);
      for (u32 t = 0; t < T; t++)
        s7 ? pass.pf_stage_twiddles<true, 7>(twl.data(), t) : strided ? pass.pf_stage_twiddles<true>(twl.data(), t) : pass.pf_stage_twiddles<false>(twl.data(), t);
      for (u32 z = p.z_base; z < p.z_base + nz; z++)
        for (u32 c = 0; c < cols; c++)
          for (u32 w = 0; w < wgs; w++) {
            for (u32 t = 0; t < T; t++) pass.prefetch(t, T, w, c, z, &regs[16 * t]);
            for (u32 t = 0; t < T; t++) {
              if (!strided) pass.pf_first_step<false, 0>(lds.data(), t, w, z, &regs[16 * t], p.group_tw);
              else if (s7 && fmode == 0) pass.pf_first_step<true, 0, 7>(lds.data(), t, w, z, &regs[16 * t], twl.data());
              else if (s7 && fmode == 1) pass.pf_first_step<true, 1, 7>(lds.data(), t, w, z, &regs[16 * t], twl.data());
              else if (s7) pass.pf_first_step<true, 2, 7>(lds.data(), t, w, z, &regs[16 * t], twl.data());
              else if (fmode == 0) pass.pf_first_step<true, 0>(lds.data(), t, w, z, &regs[16 * t], twl.data());
              else if (fmode == 1) pass.pf_first_step<true, 1>(lds.data(), t, w, z, &regs[16 * t], twl.data());
              else pass.pf_first_step<true, 2>(lds.data(), t, w, z, &regs[16 * t], twl.data());
            }
            if (s7) {
              for (u32 t = 0; t < T; t++) pass.pf_last_strided_read<7>(lds.data(), t, &regs[16 * t]);
              for (u32 t = 0; t < T; t++) pass.pf_last_strided_store<7>(t, w, c, z, &regs[16 * t]);
            } else if (strided) {
              for (u32 t = 0; t < T; t++) pass.pf_mid_step<7>(lds.data(), t, 5, twl.data() + 512);
              for (u32 t = 0; t < T; t++) pass.pf_last_strided_read(lds.data(), t, &regs[16 * t]);
              for (u32 t = 0; t < T; t++) pass.pf_last_strided_store(t, w, c, z, &regs[16 * t]);
            } else {
              for (u32 t = 0; t < T; t++) pass.pf_mid_step<7>(lds.data(), t, 9, twl.data());
              for (u32 t = 0; t < T; t++) pass.pf_mid_step<4>(lds.data(), t, 6, twl.data() + 1024);
              for (u32 t = 0; t < T; t++) pass.pf_last_contiguous(lds.data(), t);
              for (u32 t = 0; t < T; t++) pass.pf_store_wave_rows(lds.data(), t, w, c, z);
            }
          }
      return;
    }
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
  void launch_bitrev_small(const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols, unsigned long long *noncanonical) {
    for (u32 c = 0; c < cols; c++)
      for (u32 i = 0; i < (1u << lg); i++) {
        if (noncanonical && in[c * is + i] >= GL_P) *noncanonical = 1;
        out[c * os + bitrev32(i, lg)] = gl_canon(in[c * is + i]);
      }
  }
};

static u64 g_rc[POS_RC_WORDS];
static bool g_rc_ok = false;
static void rc_init() {
  if (!g_rc_ok) { pos_derive_round_constants(g_rc); pos_extend_round_constants(g_rc); g_rc_ok = true; }
}

extern "C" {
// the partial rounds three at a time (what the gfx950 form of the permutation computes), in portable arithmetic
void emu_poseidon_permute_grouped(u64 *s) {
  rc_init();
  pos_permute_grouped_portable(s, g_rc);
}
// the largest entry of the constant tables of the grouped partial rounds
u32 emu_poseidon_partial_max_entry() { return pos_partial_max_entry(); }
void emu_poseidon_permute(u64 *s) {
  rc_init();
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
// returns the number of passes that ran as the prefetching kernel (k_ntt_pass_pf)
int emu_ntt_forward(const u64 *in, u64 *out, u32 lg, u32 ncols, u64 shift, u32 zbits) {
  EmuBackend be; NttHost<EmuBackend> h(be);
  if (zbits & 0x100) { h.computed_scale = true; zbits &= 0xff; }  // test hook: the computed coset scale (FMODE 2)
  else h.computed_scale = false;
  h.forward(in, (u64)1 << lg, out, (u64)1 << (lg + zbits), lg, ncols, shift, zbits);
  return be.pf_launches;
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

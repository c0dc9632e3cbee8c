// Host-side orchestration of the multi-pass NTT: table construction (tiny, done
// once per size on the host and uploaded) and the pass schedule.  Templated on a
// backend so that the HIP library and the CPU emulation harness (tests/emu) run
// the same schedule:
//   const u64* Backend::table(const std::string& key, std::function<std::vector<u64>()> make)
//   void Backend::launch_pass(bool inverse, const NttPassParams&, u32 wgs, u32 cols, u32 z)
//   void Backend::launch_bitrev(const BitrevTile&, u32 wgs, u32 cols)      (lg >= 12)
//   void Backend::launch_bitrev_small(in, out, strides, lg, cols, noncanonical flag)   (lg < 12)
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "ntt.hpp"

namespace lcp2 {

inline std::vector<u64> make_pow_table(u64 base, u64 first, size_t count, u64 scalar = 1) {
  // out[j] = scalar * base^(first * j)
  std::vector<u64> t(count);
  u64 step = gl_pow(base, first), cur = scalar;
  for (size_t j = 0; j < count; j++) { t[j] = cur; cur = gl_mul(cur, step); }
  return t;
}

template <class Backend>
struct NttHost {
  Backend &be;
  bool computed_scale = NTT_PF_COMPUTED_SCALE;  // coset scale of the prefetching kernel computed instead of read (ntt.hpp pf_first_step)
  explicit NttHost(Backend &b) : be(b) {}

  static std::string key(const char *what, u64 a, u64 b, u64 c = 0) {
    return std::string(what) + ":" + std::to_string(a) + ":" + std::to_string(b) + ":" + std::to_string(c);
  }

  // everything a pass needs for its register steps: w_{2^B}^e table, the w_16^k constants and the step plan
  void set_group(NttPassParams &p, bool inv) {
    const u32 B = p.B;
    p.group_tw = be.table(key("group", B, inv), [=] {
      u64 w = gl_root_of_unity(B);
      if (inv) w = gl_inv(w);
      return make_pow_table(w, 1, (size_t)1 << B);
    });
    p.nsteps = ntt_step_plan(p.B, p.S, p.step_plan);
  }
  // inter-group twiddles w_{2^(g_lo + B)}^(l * bitrev(t')) of a strided pass as ONE table indexed [t'][l]: a single coalesced
  // load per element instead of two lookups and a multiply (the table is shared by every column and coset: cache resident)
  TwoLevelTable group_twiddle_table(u32 g_lo, u32 B, bool inv) {
    const u32 lg = g_lo + B;
    if (lg > NTT_DIRECT_MAX_LG) return root_table(lg, inv);
    TwoLevelTable t;
    t.h = NTT_DIRECT;
    t.hi = nullptr;
    t.lo = be.table(key("grouptw", g_lo, B, inv), [=] {
      u64 w = gl_root_of_unity(lg);
      if (inv) w = gl_inv(w);
      std::vector<u64> out((size_t)1 << lg);
      for (u32 tp = 0; tp < (1u << B); tp++) {
        const u64 step = gl_pow(w, bitrev32(tp, B));
        u64 cur = 1;
        u64 *row = out.data() + ((size_t)tp << g_lo);
        for (u64 l = 0; l < (1ull << g_lo); l++) { row[l] = cur; cur = gl_mul(cur, step); }
      }
      return out;
    });
    return t;
  }
  TwoLevelTable root_table(u32 lg, bool inv) {
    TwoLevelTable t;
    t.h = (lg + 1) / 2;
    u32 h = t.h;
    t.lo = be.table(key("rootlo", lg, inv), [=] {
      u64 w = gl_root_of_unity(lg);
      if (inv) w = gl_inv(w);
      return make_pow_table(w, 1, (size_t)1 << h);
    });
    t.hi = be.table(key("roothi", lg, inv), [=] {
      u64 w = gl_root_of_unity(lg);
      if (inv) w = gl_inv(w);
      return make_pow_table(w, 1ull << h, (size_t)1 << (lg - h));
    });
    return t;
  }
  // shift_z^j tables for z = 0..nz-1, shift_z = base * w_{2^(lg+zbits)}^z  (z = 0 only when zbits = 0);
  // inverse: (shift^-1)^j scaled by `scalar`.
  TwoLevelTable shift_table(u64 base, u32 lg, u32 zbits, bool inv, u64 scalar, u64 &lo_stride, u64 &hi_stride, bool allow_direct = false) {
    TwoLevelTable t;
    if (allow_direct && lg <= NTT_DIRECT_MAX_LG && lg >= 14) {  // one row of 2^lg powers per coset: one coalesced load, no multiply
      const u32 nzd = 1u << zbits;
      t.h = NTT_DIRECT;
      t.hi = nullptr;
      lo_stride = (u64)1 << lg;
      hi_stride = 0;
      t.lo = be.table(key("shdirect", base, lg * 64 + zbits * 2 + inv, scalar), [=] {
        std::vector<u64> all;
        all.reserve((size_t)nzd << lg);
        for (u32 z = 0; z < nzd; z++) {
          u64 sft = base;
          if (zbits) sft = gl_mul(sft, gl_pow(gl_root_of_unity(lg + zbits), z));
          if (inv) sft = gl_inv(sft);
          auto v = make_pow_table(sft, 1, (size_t)1 << lg, scalar);
          all.insert(all.end(), v.begin(), v.end());
        }
        return all;
      });
      return t;
    }
    t.h = (lg + 1) / 2;
    u32 h = t.h;
    u32 nz = 1u << zbits;
    lo_stride = (u64)1 << h;
    hi_stride = (u64)1 << (lg - h);
    auto shift_of = [=](u32 z) {
      u64 s = base;
      if (zbits) s = gl_mul(s, gl_pow(gl_root_of_unity(lg + zbits), z));
      return inv ? gl_inv(s) : s;
    };
    t.lo = be.table(key("shlo", base, lg * 64 + zbits * 2 + inv, scalar), [=] {
      std::vector<u64> all;
      for (u32 z = 0; z < nz; z++) {
        auto v = make_pow_table(shift_of(z), 1, (size_t)1 << h);
        all.insert(all.end(), v.begin(), v.end());
      }
      return all;
    });
    t.hi = be.table(key("shhi", base, lg * 64 + zbits * 2 + inv, scalar), [=] {
      std::vector<u64> all;
      for (u32 z = 0; z < nz; z++) {
        auto v = make_pow_table(shift_of(z), 1ull << h, (size_t)1 << (lg - h), scalar);
        all.insert(all.end(), v.begin(), v.end());
      }
      return all;
    });
    return t;
  }

  // [z][16]: shift_z^(k * 2^step_lg): the ratio between the 16 elements a thread of the prefetching kernel holds (ntt.hpp pf_first_step)
  const u64 *shift_step_table(u64 base, u32 lg, u32 zbits, u32 step_lg) {
    return be.table(key("shstep", base, lg * 64 + zbits, step_lg), [=] {
      std::vector<u64> all;
      for (u32 z = 0; z < (1u << zbits); z++) {
        u64 sft = base;
        if (zbits) sft = gl_mul(sft, gl_pow(gl_root_of_unity(lg + zbits), z));
        auto v = make_pow_table(sft, 1ull << step_lg, 16);
        all.insert(all.end(), v.begin(), v.end());
      }
      return all;
    });
  }

  // Forward coset NTT, natural in -> bit-reversed out.
  //   zbits = 0: one transform of size 2^lg per column, out[col] has 2^lg elements.
  //   zbits = r: low-degree extension: 2^r coset transforms of the same 2^lg coefficients,
  //              coset z (shift * w_{2^(lg+r)}^z) lands in block bitrev(z) of out[col] (2^(lg+r) elements),
  //              i.e. out is the bit-reversed order of the size-2^(lg+r) coset NTT of the zero-padded input.
  // shift = 1 and zbits = 0 gives the plain NTT (no scaling pass).
  // block_first / block_count (a power of two): only the leaf blocks [block_first, block_first + block_count) are
  // computed and `out` holds just those, block_first at offset 0 (coset-sharded LDE of SURVEY 8e).
  void forward(const u64 *in, u64 in_col_stride, u64 *out, u64 out_col_stride, u32 lg, u32 ncols, u64 shift, u32 zbits,
               u32 block_first = 0, u32 block_count = 0) {
    if (block_count == 0) block_count = 1u << zbits;
    const bool all_blocks = block_first == 0 && block_count == (1u << zbits);
    u32 lgcount = 0;
    while ((1u << lgcount) < block_count) lgcount++;
    NttGroup groups[8];
    int ng = ntt_plan(lg, groups);
    for (int gi = 0; gi < ng; gi++) {
      const NttGroup &g = groups[gi];
      NttPassParams p{};
      bool first = gi == 0;
      p.L = g.L; p.S = g.S; p.B = g.B; p.g_lo = g.g_lo;
      set_group(p, false);
      if (g.g_lo) p.tw = group_twiddle_table(g.g_lo, g.B, false);
      u32 wgs, nz;
      if (first) {
        p.in = in; p.in_col_stride = in_col_stride; p.in_z_stride = 0;
        p.out = out; p.out_col_stride = out_col_stride; p.out_z_stride = (u64)1 << lg; p.zbits = zbits;
        if (shift != 1 || zbits) {
          p.scale_mode = 2;
          p.sc = shift_table(shift, lg, zbits, false, 1, p.sc_lo_z_stride, p.sc_hi_z_stride, true);
          if (computed_scale && p.sc.h == NTT_DIRECT && ntt_pf_strided(p)) p.sc_step = shift_step_table(shift, lg, zbits, g.g_lo + 9 - g.S);  // a thread's elements are 512 >> S runs apart
        }
        wgs = 1u << (lg - g.L);
        nz = 1u << zbits;
        if (!all_blocks) {  // one launch per leaf block: block b holds coset bitrev(b)
          p.out_block_base = block_first;
          for (u32 b = block_first; b < block_first + block_count; b++) {
            p.z_base = bitrev32(b, zbits);
            be.launch_pass(false, p, wgs, ncols, 1);
          }
          continue;
        }
      } else {
        // remaining groups: in place over the whole out column (all coset blocks are just more sub-transforms)
        p.in = out; p.in_col_stride = out_col_stride;
        p.out = out; p.out_col_stride = out_col_stride;
        p.canonical_in = 1;  // written by the pass before
        wgs = 1u << (lg + lgcount - g.L);
        nz = 1;
      }
      // first pass of a full LDE: group the cosets of a slab on one XCD when the block count allows the bijection
      p.xcd_group = first && nz > 1 && ((u64)wgs * ncols) % 8 == 0 ? 1 : 0;
      if (p.xcd_group && wgs % 8 == 0) p.xcd_group = 2;  // column before slab index: the factor rows of a slab index are shared by the columns
      be.launch_pass(false, p, wgs, ncols, nz);
    }
  }

  // Inverse coset NTT, bit-reversed in -> natural out, includes 1/n and shift^-j.
  void inverse_bitrev_in(const u64 *in, u64 in_col_stride, u64 *out, u64 out_col_stride, u32 lg, u32 ncols, u64 shift) {
    NttGroup groups[8];
    int ng = ntt_plan(lg, groups);
    u64 ninv = gl_inv(((u64)1 << lg) % GL_P);
    for (int gi = ng - 1; gi >= 0; gi--) {
      const NttGroup &g = groups[gi];
      NttPassParams p{};
      p.L = g.L; p.S = g.S; p.B = g.B; p.g_lo = g.g_lo;
      set_group(p, true);
      if (g.g_lo) p.tw = group_twiddle_table(g.g_lo, g.B, true);
      bool firstpass = gi == ng - 1, lastpass = gi == 0;
      p.in = firstpass ? in : out; p.in_col_stride = firstpass ? in_col_stride : out_col_stride;
      p.out = out; p.out_col_stride = out_col_stride;
      if (lastpass) {
        if (shift == 1) { p.scale_mode = 1; p.scale_scalar = ninv; }
        else { p.scale_mode = 2; p.sc = shift_table(shift, lg, 0, true, ninv, p.sc_lo_z_stride, p.sc_hi_z_stride); }
      }
      be.launch_pass(true, p, 1u << (lg - g.L), ncols, 1);
    }
  }

  // ifft of natural-order values -> natural-order coefficients (PolynomialValues::ifft):
  // bit-reversal permutation into `out`, then the in-place inverse passes.
  // noncanonical (nullable, device word): set to 1 if an input value is >= p
  void inverse_natural(const u64 *in, u64 in_col_stride, u64 *out, u64 out_col_stride, u32 lg, u32 ncols, unsigned long long *noncanonical = nullptr) {
    if (lg >= 12) {
      BitrevTile b{in, out, in_col_stride, out_col_stride, lg, noncanonical};
      be.launch_bitrev(b, 1u << (lg - 12), ncols);
    } else {
      be.launch_bitrev_small(in, in_col_stride, out, out_col_stride, lg, ncols, noncanonical);
    }
    inverse_bitrev_in(out, out_col_stride, out, out_col_stride, lg, ncols, 1);
  }
};

}  // namespace lcp2

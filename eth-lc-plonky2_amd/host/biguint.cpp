// BigUint gadgets (biguint.hpp).
#include "biguint.hpp"
#include "gadgets.hpp"
#include "host_internal.hpp"

namespace lc {

BigUintValue biguint_from_u64(uint64_t v) {
  BigUintValue r{(uint32_t)v, (uint32_t)(v >> 32)};
  while (r.size() > 1 && r.back() == 0) r.pop_back();
  return r;
}

namespace {
U32Target limb_or_zero(CircuitBuilder &b, const BigUintTarget &x, size_t i) { return i < x.num_limbs() ? x.limbs[i] : b.zero_u32(); }

// x < 2^33 as (low 32 bits, bit 32)
std::pair<Target, BoolTarget> split_33(CircuitBuilder &b, Target x) {
  std::vector<BoolTarget> bits = b.split_le(x, 33);
  return {b.le_sum(bits, 0, 32), bits[32]};
}
// a * b + c of three 32-bit words (< p: at most 2^64 - 2^32) as canonical (low, high) words
std::pair<U32Target, U32Target> mul_add_u32(CircuitBuilder &b, U32Target x, U32Target y, U32Target z) {
  CircuitBuilder::CanonicalSplit s = b.split_canonical(b.mul_add(x.t, y.t, z.t));
  return {U32Target{s.lo}, U32Target{s.hi}};
}

// host arithmetic on limb vectors for the division generator
BigUintValue trimmed(BigUintValue v) { while (v.size() > 1 && v.back() == 0) v.pop_back(); if (v.empty()) v.push_back(0); return v; }
int cmp_value(const BigUintValue &a_, const BigUintValue &b_) {
  BigUintValue a = trimmed(a_), b = trimmed(b_);
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
void sub_in_place(BigUintValue &a, const BigUintValue &b) {  // a >= b
  uint64_t borrow = 0;
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t d = (uint64_t)a[i] - (i < b.size() ? b[i] : 0) - borrow;
    a[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
}
// schoolbook binary long division: fine for the handful of divisions a witness needs
void div_rem_value(const BigUintValue &a, const BigUintValue &b, BigUintValue &q, BigUintValue &r) {
  q.assign(a.size(), 0);
  r.assign(b.size() + 1, 0);
  for (size_t bit = a.size() * 32; bit-- > 0;) {
    uint32_t carry = (a[bit / 32] >> (bit % 32)) & 1;  // r = 2 r + bit
    for (size_t i = 0; i < r.size(); i++) { uint32_t nc = r[i] >> 31; r[i] = (r[i] << 1) | carry; carry = nc; }
    if (cmp_value(r, b) >= 0) { sub_in_place(r, b); q[bit / 32] |= 1u << (bit % 32); }
  }
  r.resize(b.size());
}
}  // namespace

BigUintTarget add_virtual_biguint_target(CircuitBuilder &b, size_t num_limbs) {
  BigUintTarget t;
  for (size_t i = 0; i < num_limbs; i++) {
    t.limbs.push_back(U32Target{b.add_virtual_target()});
    b.split_le(t.limbs.back().t, 32);
  }
  return t;
}
BigUintTarget constant_biguint(CircuitBuilder &b, const BigUintValue &value) {
  BigUintTarget t;
  for (uint32_t l : value) t.limbs.push_back(b.constant_u32(l));
  return t;
}
void connect_biguint(CircuitBuilder &b, const BigUintTarget &lhs, const BigUintTarget &rhs) {
  const size_t n = std::max(lhs.num_limbs(), rhs.num_limbs());
  for (size_t i = 0; i < n; i++) b.connect_u32(limb_or_zero(b, lhs, i), limb_or_zero(b, rhs, i));
}
void set_biguint_target(PartialWitness &pw, const BigUintTarget &t, const BigUintValue &value) {
  for (size_t i = 0; i < t.num_limbs(); i++) pw.set_u32_target(t.limbs[i], i < value.size() ? value[i] : 0);
  for (size_t i = t.num_limbs(); i < value.size(); i++)
    if (value[i]) throw std::runtime_error("set_biguint_target: the value does not fit the target's limbs");
}

BigUintTarget add_biguint(CircuitBuilder &b, const BigUintTarget &x, const BigUintTarget &y) {
  const size_t n = std::max(x.num_limbs(), y.num_limbs());
  BigUintTarget r;
  Target carry = b.zero();
  for (size_t i = 0; i < n; i++) {
    Target s = b.add(b.add(limb_or_zero(b, x, i).t, limb_or_zero(b, y, i).t), carry);  // < 2^33
    auto [low, c] = split_33(b, s);
    r.limbs.push_back(U32Target{low});
    carry = c.target;
  }
  r.limbs.push_back(U32Target{carry});
  return r;
}

BigUintTarget mul_biguint(CircuitBuilder &b, const BigUintTarget &x, const BigUintTarget &y) {
  const size_t nx = x.num_limbs(), ny = y.num_limbs();
  std::vector<U32Target> acc(nx + ny, b.zero_u32());
  for (size_t i = 0; i < nx; i++) {
    Target carry = b.zero();  // < 2^32 throughout: x_i y_j + acc + carry <= 2^64 - 1
    for (size_t j = 0; j < ny; j++) {
      auto [lo, hi] = mul_add_u32(b, x.limbs[i], y.limbs[j], acc[i + j]);
      auto [low, c] = split_33(b, b.add(lo.t, carry));
      acc[i + j] = U32Target{low};
      carry = b.add(hi.t, c.target);
    }
    acc[i + ny] = U32Target{carry};  // untouched so far in this and earlier rows beyond their own carry slot
  }
  return BigUintTarget{acc};
}

BoolTarget cmp_biguint(CircuitBuilder &b, const BigUintTarget &x, const BigUintTarget &y) {
  // y - x limb by limb with a borrow: d_i = 2^32 + y_i - x_i - borrow in [1, 2^33); bit 32 set = no borrow out
  const size_t n = std::max(x.num_limbs(), y.num_limbs());
  Target borrow = b.zero();
  for (size_t i = 0; i < n; i++) {
    Target d = b.sub(b.add_const(b.sub(limb_or_zero(b, y, i).t, limb_or_zero(b, x, i).t), 1ull << 32), borrow);
    borrow = b.not_(split_33(b, d).second).target;
  }
  return b.not_(BoolTarget{borrow});  // no borrow at the top: x <= y
}

std::pair<BigUintTarget, BigUintTarget> div_rem_biguint(CircuitBuilder &b, const BigUintTarget &x, const BigUintTarget &y) {
  const size_t nx = x.num_limbs(), ny = y.num_limbs();
  std::vector<Target> in;
  for (auto &l : x.limbs) in.push_back(l.t);
  for (auto &l : y.limbs) in.push_back(l.t);
  std::vector<Target> out = b.hint(in, nx + ny, [nx, ny](const std::vector<F> &v, std::vector<F> &o) {
    BigUintValue a(nx), d(ny), q, r;
    for (size_t i = 0; i < nx; i++) a[i] = (uint32_t)v[i];
    for (size_t i = 0; i < ny; i++) d[i] = (uint32_t)v[nx + i];
    if (cmp_value(d, {0}) == 0) throw UnsatisfiedError("div_rem_biguint: division by zero");
    div_rem_value(a, d, q, r);
    for (size_t i = 0; i < nx; i++) o[i] = q[i];
    for (size_t i = 0; i < ny; i++) o[nx + i] = r[i];
  });
  BigUintTarget q, r;
  for (size_t i = 0; i < nx; i++) { q.limbs.push_back(U32Target{out[i]}); b.split_le(out[i], 32); }
  for (size_t i = 0; i < ny; i++) { r.limbs.push_back(U32Target{out[nx + i]}); b.split_le(out[nx + i], 32); }
  connect_biguint(b, add_biguint(b, mul_biguint(b, q, y), r), x);  // x = q y + r
  b.connect(cmp_biguint(b, y, r).target, b.zero());                // not (y <= r): r < y
  return {q, r};
}

IsEqualBigUint add_virtual_is_equal_big_uint_target(CircuitBuilder &b) {
  IsEqualBigUint t{add_virtual_biguint_target(b, 8), add_virtual_biguint_target(b, 8), BoolTarget{b.one()}};
  for (size_t i = 0; i < 8; i++) t.result = b.and_(t.result, b.is_equal(t.big1.limbs[i].t, t.big2.limbs[i].t));
  return t;
}

BigUintHash256ConnectTarget add_virtual_biguint_hash256_connect_target_big(CircuitBuilder &b) {
  BigUintHash256ConnectTarget t{add_virtual_biguint_target(b, 8), b.add_virtual_hash256_target()};
  for (size_t i = 0; i < 8; i++) {  // limb i of the integer = the bytes of hash word i in reverse order (src/utils.rs:100-110)
    std::vector<BoolTarget> big = b.split_le(t.big.limbs[i].t, 32), h = b.split_le(t.h256[i].t, 32);
    for (size_t k = 0; k < 4; k++)
      for (size_t j = 0; j < 8; j++) b.connect(big[8 * k + j].target, h[24 - 8 * k + j].target);
  }
  return t;
}

FindSyncCommitteeBigTarget add_virtual_find_sync_committee_target_big(CircuitBuilder &b) {
  FindSyncCommitteeBigTarget t;
  t.attested_slot_big = add_virtual_biguint_target(b, 8);
  t.cur_slot_big = add_virtual_biguint_target(b, 8);
  t.cur_sync_committee_i = b.add_virtual_hash256_target();
  t.cur_sync_committee_ii = b.add_virtual_hash256_target();
  t.sync_committee_for_attested_slot = b.add_virtual_hash256_target();
  const BigUintTarget one_big = constant_biguint(b, {1}), n_slot = constant_biguint(b, biguint_from_u64(N_SLOTS_PER_PERIOD));
  const BigUintTarget attested_period = div_rem_biguint(b, t.attested_slot_big, n_slot).first;
  const BigUintTarget cur_period = div_rem_biguint(b, t.cur_slot_big, n_slot).first;
  const BigUintTarget next_period = add_biguint(b, cur_period, one_big);
  IsEqualBigUint from_cur = add_virtual_is_equal_big_uint_target(b);
  connect_biguint(b, from_cur.big1, attested_period);
  connect_biguint(b, from_cur.big2, cur_period);
  IsEqualBigUint from_next = add_virtual_is_equal_big_uint_target(b);
  connect_biguint(b, from_next.big1, attested_period);
  connect_biguint(b, from_next.big2, next_period);
  b.connect(b.or_(from_cur.result, from_next.result).target, b.one());  // the attested slot is from the current or the next period
  for (size_t i = 0; i < 8; i++)
    b.connect(t.sync_committee_for_attested_slot[i].t, b.select(from_cur.result, t.cur_sync_committee_i[i].t, t.cur_sync_committee_ii[i].t));
  t.is_attested_from_next_period = from_next.result;
  return t;
}

UpdateValidityBigTarget add_virtual_update_validity_target_big(CircuitBuilder &b) {
  UpdateValidityBigTarget t{add_virtual_biguint_target(b, 8), add_virtual_biguint_target(b, 8), add_virtual_biguint_target(b, 1)};
  b.connect(cmp_biguint(b, t.cur_slot_big, t.finalized_slot_big).target, b.one());  // cur_slot <= finalized_slot
  b.connect(cmp_biguint(b, t.participation_big, constant_biguint(b, biguint_from_u64(FINALITY_THRESHOLD))).target, b.zero());  // participation > threshold
  return t;
}

}  // namespace lc

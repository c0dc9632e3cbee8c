// BigUint gadgets: what the reference uses of plonky2_crypto's biguint module (CircuitBuilderBiguint, WitnessBigUint;
// src/targets.rs:184-235, 304-332, src/utils.rs:76-113), with the same names and argument order:
//   builder.add_virtual_biguint_target(n), constant_biguint, connect_biguint, add_biguint, mul_biguint, div_rem_biguint,
//   cmp_biguint (a <= b), witness.set_biguint_target, and the reference's own add_virtual_is_equal_big_uint_target.
// A BigUintTarget is little-endian 32-bit limbs.  plonky2_crypto builds the arithmetic from U32 gates (U32ArithmeticGate,
// U32AddManyGate, ComparisonGate), whose source is not visible from the reference; here every limb operation is ArithmeticGate
// operations plus bit decompositions of the host layer: a limb product a * b + c < p is split canonically into low and high
// word, sums and differences carry through a 33-bit decomposition.  Same values, same constraints on them, more rows.
#pragma once
#include "lc_plonky2.hpp"

namespace lc {

using BigUintValue = std::vector<uint32_t>;  // little-endian limbs
BigUintValue biguint_from_u64(uint64_t v);

struct BigUintTarget {
  std::vector<U32Target> limbs;
  size_t num_limbs() const { return limbs.size(); }
  U32Target get_limb(size_t i) const { return limbs[i]; }
};

// limbs are range checked on creation (plonky2_crypto leaves that to the gates that consume them)
BigUintTarget add_virtual_biguint_target(CircuitBuilder &builder, size_t num_limbs);
BigUintTarget constant_biguint(CircuitBuilder &builder, const BigUintValue &value);
void connect_biguint(CircuitBuilder &builder, const BigUintTarget &lhs, const BigUintTarget &rhs);  // the shorter one is zero-extended
BigUintTarget add_biguint(CircuitBuilder &builder, const BigUintTarget &a, const BigUintTarget &b);
BigUintTarget mul_biguint(CircuitBuilder &builder, const BigUintTarget &a, const BigUintTarget &b);
// (a / b, a % b); b = 0 cannot be proved
std::pair<BigUintTarget, BigUintTarget> div_rem_biguint(CircuitBuilder &builder, const BigUintTarget &a, const BigUintTarget &b);
BoolTarget cmp_biguint(CircuitBuilder &builder, const BigUintTarget &a, const BigUintTarget &b);  // a <= b
void set_biguint_target(PartialWitness &witness, const BigUintTarget &target, const BigUintValue &value);

// src/utils.rs:76-90
struct IsEqualBigUint { BigUintTarget big1, big2; BoolTarget result; };
IsEqualBigUint add_virtual_is_equal_big_uint_target(CircuitBuilder &builder);
// src/utils.rs:93-113: a BigUintTarget of 8 limbs tied bit by bit to the Hash256Target that holds its little-endian bytes
struct BigUintHash256ConnectTarget { BigUintTarget big; Hash256Target h256; };
BigUintHash256ConnectTarget add_virtual_biguint_hash256_connect_target_big(CircuitBuilder &builder);

// src/targets.rs:184-235 and :304-332 in the reference's BigUint form (gadgets.hpp has the u64 form the light-client circuit uses)
struct FindSyncCommitteeBigTarget {
  BigUintTarget attested_slot_big, cur_slot_big;
  Hash256Target cur_sync_committee_i, cur_sync_committee_ii, sync_committee_for_attested_slot;
  BoolTarget is_attested_from_next_period;
};
FindSyncCommitteeBigTarget add_virtual_find_sync_committee_target_big(CircuitBuilder &builder);
struct UpdateValidityBigTarget { BigUintTarget cur_slot_big, finalized_slot_big, participation_big; };
UpdateValidityBigTarget add_virtual_update_validity_target_big(CircuitBuilder &builder);

}  // namespace lc

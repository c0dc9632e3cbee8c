#!/bin/bash
# The oracle (the checker of every parity test) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: builds oracle/*.c
# into /tmp/liboracle_asan.so and runs the oracle-only tests against it (golden vectors, Poseidon / NTT / field checks, small
# proofs through the oracle's prover and both verifiers).  GPU sanitizers are not available on the pool; this is the CPU half.
set -e
cd "$(dirname "$0")/../.."
gcc -O1 -g -fPIC -shared -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -o /tmp/liboracle_asan.so oracle/*.c -lm
export LCP2_ORACLE_LIB=/tmp/liboracle_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
python -m pytest tests/test_oracle_golden.py tests/test_host_verifier.py tests/test_emu_kernels.py -x -q -m "not gpu" "$@"
# the C++ host layer (CircuitBuilder, gadgets, recursive verifier, BigUint) under the same sanitizers: the circuit-level tests in cpu mode
unset LD_PRELOAD
H=eth-lc-plonky2_amd/host
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -o /tmp/test_gadgets_asan tests/cpp/test_gadgets.cpp \
    $H/gates.cpp $H/builder.cpp $H/gadgets.cpp $H/light_client_update.cpp $H/poseidon_host.cpp $H/recursion.cpp $H/biguint.cpp \
    -L eth-lc-plonky2_amd -llcp2 -L oracle -loracle -Wl,-rpath,$PWD/eth-lc-plonky2_amd -Wl,-rpath,$PWD/oracle -fopenmp
for t in test_merkle_root_4_leaves test_contract_state test_builder_primitives test_biguint_arithmetic_all_ones test_find_sync_committee_big_next_period \
         test_recursive_verifier test_recursive_verifier_tampered_leaf_panics test_update_validity_big_threshold_not_exceeded_panics; do
  /tmp/test_gadgets_asan cpu $t | tail -1
done

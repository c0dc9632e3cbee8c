#!/bin/bash
# Whole-proof parity (GPU proof == CPU oracle proof, word for word) of every workload bench.py times, at the workload's own size.
# The oracle needs minutes per circuit at these sizes, so this is not part of the test-suite: run one item per gpurun call
# (each stays below the 20-minute limit of a call on the box's 32 host threads) and keep the outputs under profiles/.
#   bash tests/checks/parity_at_size.sh synthetic-2p22 | mix-2p22 | mix-2p20 | lc-step | lc-step-recursive | headline
# A heartbeat line every minute keeps the box from taking the silent oracle for hung.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 2
mkdir -p gpurun_out
( while sleep 60; do echo "[heartbeat $(date +%T)]"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
export OMP_NUM_THREADS=32
case "$1" in
  synthetic-2p22)    python3 tests/checks/parity_full_size.py 22 > gpurun_out/parity_full_size.json ;;
  mix-2p22)          python3 tests/checks/parity_full_size.py 22 reference-mix > gpurun_out/parity_mix_2p22.json ;;
  mix-2p20)          python3 tests/checks/parity_full_size.py 20 reference-mix > gpurun_out/parity_mix_2p20.json ;;
  lc-step)           LCP2_ORACLE_PROVE_ALL=1 tests/cpp/test_gadgets gpu test_light_client_update > gpurun_out/real_gadget_parity_lc_step.log 2>&1 ;;
  lc-step-recursive) LCP2_ORACLE_PROVE_ALL=1 tests/cpp/test_gadgets gpu test_light_client_update_with_recursive_proof > gpurun_out/real_gadget_parity_recursive.log 2>&1 ;;
  headline)          LCP2_ORACLE_PROVE_ALL=1 tests/cpp/test_gadgets gpu test_real_gadget_circuit_2p22 > gpurun_out/real_gadget_parity_2p22.log 2>&1 ;;
  *) echo "usage: $0 synthetic-2p22|mix-2p22|mix-2p20|lc-step|lc-step-recursive|headline" >&2; exit 2 ;;
esac
rc=$?
echo "parity_at_size $1: rc=$rc"
exit $rc

#!/bin/bash
# rocprofv3 passes over the real light-client step (examples/lc_prover) : kernel trace, then SQ counters
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/lc_kt -- ./examples/lc_prover tmp_fixtures/u633.json tmp_fixtures/u634.json --repeat 2 > $out/lc_kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $out/lc_pmc -- ./examples/lc_prover tmp_fixtures/u633.json tmp_fixtures/u634.json --repeat 1 > $out/lc_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/syn_kt -- python3 tools/prof_prove.py 20 2 > $out/syn_kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $out/syn_pmc -- python3 tools/prof_prove.py 20 1 > $out/syn_pmc.log 2>&1

#!/bin/bash
# rocprofv3 kernel traces: the real light-client step and a 2^20-row synthetic proof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/lc_kt -- ./examples/lc_prover tmp_fixtures/u633.json tmp_fixtures/u634.json --repeat 2 > $out/lc_kt.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $out/syn_kt -- python3 tools/prof_prove.py 20 2 > $out/syn_kt.log 2>&1

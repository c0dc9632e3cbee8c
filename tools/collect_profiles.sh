#!/bin/bash
# Runs on the GPU box (gpurun): every measurement DESIGN.md / bench.py quote for one round, into $1 (under gpurun_out/).
#   rocprofv3 kernel trace + stats of bench.py (headline only: the 2^22-row real-gadget circuit, data.prove(pw) in-process), FETCH_SIZE and
#   WRITE_SIZE PMC passes of the same command (separate, as MI355X_MICROARCH.md prescribes), SQ counters of a 2^22 proof, kernel trace of the
#   light-client step (lc_prover) and its idle-time summary, the reference gate set (tools/reference_mix_probe.py) under the same two passes,
#   the multiply micro-benchmark (tools/ubench/mulchain.hip), the per-rank compute of a sharded proof, the sharded code path over a 1-rank RCCL group, the oracle's 2^20 timing sample, the plain bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1
mkdir -p $out
# the two light-client update files of the reference, out of the committed fixture (tests/golden/lc_updates.json)
if [ ! -f tmp_fixtures/u634.json ]; then
  mkdir -p tmp_fixtures
  python3 -c "import json; d = json.load(open('tests/golden/lc_updates.json')); [json.dump(d[k], open('tmp_fixtures/u%s.json' % k, 'w')) for k in ('633', '634')]" || exit 1
fi
B="python3 bench.py --no-cpu-baseline --no-real-gadgets --no-synthetic"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B --steps 2 --warmup 1 > $out/bench_under_rocprof.json 2> $out/kt.err && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- $B --steps 1 --warmup 0 > $out/bench_under_pmc_fetch.json 2> $out/fetch.err && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- $B --steps 1 --warmup 0 > $out/bench_under_pmc_write.json 2> $out/write.err && \
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o sq -- python3 tools/prof_prove.py 22 1 > $out/sq.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $out/lc -o lc -- ./examples/lc_prover tmp_fixtures/u633.json tmp_fixtures/u634.json --repeat 3 > $out/lc.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mixkt -o mixkt -- python3 tools/reference_mix_probe.py 22 2 --no-regs > $out/mix_under_rocprof.json 2> $out/mixkt.err && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/mixfetch -o mixfetch -- python3 tools/reference_mix_probe.py 22 1 --no-regs > $out/mix_under_fetch.json 2> $out/mixfetch.err && \
python3 tools/reference_mix_probe.py 22 3 > $out/mix_native.json 2> $out/mix_native.err && \
python3 tools/reference_mix_probe.py 22 2 --interpreted > $out/mix_interpreted.json 2> $out/mix_interpreted.err && \
for rw in 0/2 0/4 0/8 5/8; do python3 tools/sharded_proof_demo.py --degree-bits 22 --reps 3 --sharded-columns --row-exchange --rehearse $rw 2>&1 | grep rehearsal >> $out/sharded_rehearsal.log || exit 1; done && \
for rw in 0/2 0/4 0/8 5/8; do python3 tools/sharded_proof_demo.py --degree-bits 22 --reps 3 --sharded-columns --row-exchange --chunked --rehearse $rw 2>&1 | grep rehearsal >> $out/sharded_rehearsal_chunked.log || exit 1; done && \
python3 bench.py --force-sharded --no-cpu-baseline --no-real-gadgets --no-synthetic > $out/bench_force_sharded.json 2> $out/bench_force_sharded.err && \
python3 tests/checks/cpu_baseline_scaling.py 20 > $out/cpu_baseline_scaling.json 2> $out/cpu_baseline_scaling.err && \
python3 bench.py > $out/bench.json 2> $out/bench.err && \
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/mulchain tools/ubench/mulchain.hip > /dev/null 2>&1 && \
for w in 2 4 8; do /tmp/mulchain $w || exit 1; done > $out/ubench_mulchain.txt
echo "collect rc=$?"
find $out -name "*.csv" | head -40

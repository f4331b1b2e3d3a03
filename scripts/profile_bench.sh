#!/bin/bash
# rocprofv3 passes of bench.py on the GPU box: kernel trace + stats, then separate PMC passes (never combined with other
# trace domains). usage (from the repo root on the box): bash scripts/profile_bench.sh <tag> [extra bench.py args]
# Outputs under gpurun_out/prof_<tag>/ ; summarise with scripts/pmc_summary.py and copy into profiles/rNN/.
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--steps 4 --warmup 1 --no-cpu-baseline --no-host-callback --survey-steps 0 $*"
# the stats pass runs 33 solves: with 5 the kernel averages carry the clock ramp of a fresh process (the caller GEMM 1.39 ms
# instead of 1.33 by the run's own HIP events); the counter passes stay short
sargs="--steps 30 --warmup 3 --no-cpu-baseline --no-host-callback --survey-steps 0 $*"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/stats" -o stats -- python3 "$root/bench.py" $sargs > "$out/stats.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $c --kernel-trace -d "$out/pmc_$c" -o pmc -- python3 "$root/bench.py" $args > "$out/pmc_$c.log" 2>&1 || exit 1
done
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace -d "$out/pmc_MFMA" -o pmc \
    -- python3 "$root/bench.py" $args > "$out/pmc_MFMA.log" 2>&1 || exit 1
# SQ / LDS view of the latency-bound kernels (k_lm_solve: one workgroup): two passes of four counters each
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d "$out/pmc_SQ1" -o pmc \
    -- python3 "$root/bench.py" $args > "$out/pmc_SQ1.log" 2>&1 || echo "SQ pass 1 failed (counter set not available?)"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d "$out/pmc_SQ2" -o pmc \
    -- python3 "$root/bench.py" $args > "$out/pmc_SQ2.log" 2>&1 || echo "SQ pass 2 failed (counter set not available?)"
find "$out" -name "*.csv" | head -20

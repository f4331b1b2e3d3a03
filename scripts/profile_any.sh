#!/bin/bash
# rocprofv3 passes of an arbitrary bench.py command line on the GPU box: kernel trace + stats, then ONE counter set per pass
# (never combined with other trace domains). usage (repo root on the box):
#   bash scripts/profile_any.sh <tag> "<bench.py args>" [counter-set ...]
# counter sets: HBM (FETCH_SIZE, WRITE_SIZE: two passes)  MFMA  SQ1  SQ2  VALU  (default: HBM MFMA)
# Outputs under gpurun_out/prof_<tag>/ ; summarise with scripts/pmc_summary2.py.
set -u
tag=$1; bargs=$2; shift 2
sets=${*:-HBM MFMA}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d "$out/pmc_$name" -o pmc -- python3 "$root/bench.py" $bargs > "$out/pmc_$name.log" 2>&1 \
        || echo "pass $name failed (counter set not available?)"
}
# (rocprofv3 7.2 can crash in its own teardown AFTER the tables are written -- seen with the cooperative launch of cfg 2: go on if they exist)
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/stats" -o stats -- python3 "$root/bench.py" $bargs > "$out/stats.log" 2>&1 \
    || [ -n "$(find "$out/stats" -name '*kernel_stats.csv' 2>/dev/null)" ] || exit 1
for s in $sets; do
    case $s in
    HBM)  run FETCH_SIZE FETCH_SIZE; run WRITE_SIZE WRITE_SIZE ;;
    MFMA) run MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE ;;
    SQ1)  run SQ1 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU ;;
    SQ2)  run SQ2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU ;;
    VALU) run VALU1 SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
          run VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F32 SQ_BUSY_CU_CYCLES ;;
    esac
done
find "$out" -name "*_kernel_trace.csv" -delete     # the traces are large; the stats and counter tables are what is kept
find "$out" -name "*.csv" | head -30

#!/bin/bash
# rocprofv3 counter passes on tools/one_conv.py; usage: tools/pmc_one_conv.sh <tag> <one_conv args...>
# One hardware block per pass where the slots demand it (MI355X_MICROARCH.md, "rocprofv3 PMC slots": TCC has 4 slots,
# FETCH_SIZE takes 3, WRITE_SIZE 2), like tools/profile.sh.  A failing pass ends the script with its log tail and a
# non-zero status: a profiler abort must not read as "no data".
set -o pipefail
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
CMD="python3 tools/one_conv.py --reps 5 $*"
pass() {   # pass <n> <counters...>
    local n=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" -d $out/pmc$n -o run --output-format csv -- $CMD > $out/pmc$n.log 2>&1 \
        || { echo "pmc pass $n ($*) FAILED:"; tail -5 $out/pmc$n.log; exit 1; }
}
pass 1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA
pass 2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC
pass 3 FETCH_SIZE
pass 4 WRITE_SIZE
pass 5 TCC_HIT_sum TCC_MISS_sum
python3 tools/summarize_prof.py $out > /dev/null || { echo "summarize_prof.py failed"; exit 1; }
grep -A 24 "^conv_bf16\|^conv_f32_mfma" $out/summary_pmc.txt | head -${LINES_OUT:-60}

#!/bin/bash
# rocprofv3 counter passes on tools/one_conv.py; usage: tools/pmc_one_conv.sh <tag> <one_conv args...>
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
CMD="python3 tools/one_conv.py --reps 5 $*"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA -d $out/pmc1 -o run --output-format csv -- $CMD > $out/pmc1.log 2>&1 || tail -3 $out/pmc1.log
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -d $out/pmc2 -o run --output-format csv -- $CMD > $out/pmc2.log 2>&1 || tail -3 $out/pmc2.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc3 -o run --output-format csv -- $CMD > $out/pmc3.log 2>&1 || tail -3 $out/pmc3.log
python3 tools/summarize_prof.py $out > /dev/null 2>&1
grep -A 24 "^conv_bf16\|^conv_f32_mfma" $out/summary_pmc.txt | head -${LINES_OUT:-60}

#!/bin/bash
# rocprofv3 passes over a short bench run; writes raw output under gpurun_out/prof_<tag>/ and compact summaries
# under gpurun_out/prof_<tag>/summary_*.txt (copy those into profiles/).
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r01}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-sclk --parity-images 0 $*"   # no parity gate: the bf16 gate runs one extra forward that no NMS launch counts   # eager launches only: with --graph bench.py times the conv stack in extra forwards that no NMS launch counts
echo "== kernel trace / stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- $BENCH > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
echo "== pmc pass 1 (SQ/GRBM)"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 \
  -d $out/pmc1 -o run --output-format csv -- $BENCH > $out/pmc1.log 2>&1 || { tail -5 $out/pmc1.log; exit 1; }
echo "== pmc pass 2 (LDS/VMEM)"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA \
  -d $out/pmc2 -o run --output-format csv -- $BENCH > $out/pmc2.log 2>&1 || { tail -5 $out/pmc2.log; exit 1; }
echo "== pmc pass 3 (FETCH_SIZE)"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o run --output-format csv -- $BENCH > $out/pmc3.log 2>&1 || { tail -5 $out/pmc3.log; }
echo "== pmc pass 4 (WRITE_SIZE)"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o run --output-format csv -- $BENCH > $out/pmc4.log 2>&1 || { tail -5 $out/pmc4.log; }
echo "== pmc pass 5 (L2 hit/miss)"
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/pmc5 -o run --output-format csv -- $BENCH > $out/pmc5.log 2>&1 || { tail -5 $out/pmc5.log; }
python3 tools/summarize_prof.py $out

#!/bin/bash
# Hardware counters of one probe command, as MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes (SQ set,
# then FETCH_SIZE + L2 hit/miss, then WRITE_SIZE), no trace domain beside --pmc, the program directly after `--`.
#   bash tools/collect_pmc.sh <tag> python3 tools/op_probe.py linear_ex --M 65536 --N 3072 --K 768 --epi 1
# -> gpurun_out/pmc_<tag>/{sq,fetch,write}/p_counter_collection.csv ; tools/summarize_pmc.py <tag> reduces them.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- "$@" > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -o p -- "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -o p -- "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- "$@" > $OUT/trace.log 2>&1
find $OUT -name "*.csv" | head -20

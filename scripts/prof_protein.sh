#!/bin/bash
# counters of the protein kernel, one pass of 2 M alignments (run on the GPU box from the repo root): bash scripts/prof_protein.sh NAME
NAME=${1:-prot}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
run() { local label=$1; shift; timeout -k 10 280 rocprofv3 "$@" > $O/$label.log 2>&1 || echo "$label failed" | tee -a $O/failed.txt; }
PM="--output-format csv"
P="python3 scripts/protein_bench.py --steps 1 --check 0"
run prot_trace --kernel-trace --stats $PM -d $O/prot_trace -- $P
run prot_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES $PM -d $O/prot_sq -- $P
run prot_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/prot_lds -- $P
if [ -z "$SHORT" ]; then
run prot_fetch --pmc FETCH_SIZE $PM -d $O/prot_fetch -- $P
run prot_write --pmc WRITE_SIZE $PM -d $O/prot_write -- $P
fi
python3 scripts/summarize_prof.py $(ls -d $O/prot_*/) > $O/protein_summary.txt 2>&1
cat $O/failed.txt 2>/dev/null

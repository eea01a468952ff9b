#!/bin/bash
# rocprofv3 passes of the bench path (run on the GPU box from the repo root): kernel trace, then one PMC pass per counter set
# (FETCH_SIZE and WRITE_SIZE in separate passes: together they abort on this ROCm).  Usage: bash scripts/prof_headline.sh NAME [passes]
set -e
NAME=${1:-prof}; PASSES=${2:-"trace sq fetch write lds"}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --pairs 1000000"
for p in $PASSES; do
  echo "pass $p"
  case $p in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-secondary > $O.trace.log 2>&1 ;;
    sq)    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- $B > $O.s.log 2>&1 ;;
    fetch) timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O.f.log 2>&1 ;;
    write) timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O.w.log 2>&1 ;;
    lds)   timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $O/pmc_lds -- $B > $O.l.log 2>&1 ;;
  esac
done
python3 scripts/summarize_prof.py $(ls -d $O/*/) > $O/summary.txt
tail -60 $O/summary.txt

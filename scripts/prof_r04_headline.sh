#!/bin/bash
# round 4: the kernel trace of the headline bench command (what bench.py's kernel_ms must agree with) and the LDS-conflict pass of the
# headline kernel (it now hands its results over through LDS).  bash scripts/prof_r04_headline.sh NAME
NAME=${1:-r04_headline}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary --no-extra > $O/bench_under_trace.log 2>&1 || echo "trace failed"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --workspace-gib 8 --pairs 2097152"
timeout -k 10 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/lane256_lds -- $B > $O/lane256_lds.log 2>&1 || echo "lds failed"
python3 scripts/summarize_prof.py $O/trace/ $O/lane256_lds/ > $O/summary.txt
tail -1 $O/bench_under_trace.log >> $O/summary.txt
cat $O/summary.txt | cut -c1-400

#!/bin/bash
# rocprofv3 kernel trace of the headline bench command (run on the GPU box from the repo root): the per-kernel average
# durations that bench.py's roofline object must agree with.  Usage: bash scripts/bench_trace.sh NAME
set -e
NAME=${1:-bench_trace}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary --no-extra > $O.log 2>&1
python3 scripts/summarize_prof.py $O/trace/ > $O/summary.txt
tail -1 $O.log >> $O/summary.txt
cat $O/summary.txt

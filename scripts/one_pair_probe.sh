#!/bin/bash
# one pair per call from native threads (tests/cpp/coalesce_bench THREADS CALLS WAIT_US): through the mailbox service, through the
# coalescer alone (MGL_SW_SERVICE_SLOTS=0) with its own accounting, and under the kernel trace.  Usage: bash scripts/one_pair_probe.sh NAME
set -e
NAME=${1:-one_pair}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for t in 1 4 16 32 64; do timeout -k 10 120 tests/cpp/coalesce_bench $t 3000 50; done > $O/mailboxes.txt 2>&1
for t in 1 4 16 32 64; do MGL_SW_SERVICE_SLOTS=0 timeout -k 10 120 tests/cpp/coalesce_bench $t 3000 50; done > $O/coalescer.txt 2>&1
MGL_SW_SERVICE_SLOTS=0 MGL_SW_DEBUG_COALESCE_TIMING=1 timeout -k 10 120 tests/cpp/coalesce_bench 16 3000 50 > $O/timing.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- tests/cpp/coalesce_bench 16 3000 50 > $O/trace.log 2>&1 || echo "trace failed"
python3 scripts/summarize_prof.py $O/trace/ > $O/summary.txt || true
cat $O/mailboxes.txt $O/coalescer.txt $O/timing.txt $O/summary.txt

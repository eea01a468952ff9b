#!/bin/bash
# Round-2 profiles of the headline path (run on the GPU box from the repo root): kernel trace of the bench command, then
# one PMC pass per counter set on a 2 M-pair launch of the same kernel.  Usage: bash scripts/prof_r02.sh NAME
set -e
NAME=${1:-r02}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary --no-extra > $O.trace.log 2>&1 || echo "trace pass failed"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --pairs 2097152"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O.s.log 2>&1 || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O.f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O.w.log 2>&1 || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $B > $O.t.log 2>&1 || echo "tcc pass failed"
python3 scripts/summarize_prof.py $(ls -d $O/*/) > $O/summary.txt
tail -5 $O.trace.log
cat $O/summary.txt

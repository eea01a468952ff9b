#!/bin/bash
# Long-read path (BASELINE.json configs[3]): kernel trace of scripts/long_read_bench.py, then PMC passes (instruction mix,
# HBM bytes) on one pass of the same batch.  Run on the GPU box from the repo root: [WS=GiB] bash scripts/prof_long.sh NAME [PAIRS]
set -e
NAME=${1:-r02_long}; PAIRS=${2:-1024}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
B="python3 scripts/long_read_bench.py $PAIRS ${WS:-120} 10000 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --seconds 3 > $O.trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O.s.log 2>&1 || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O.f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O.w.log 2>&1 || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_misc -- $B > $O.t.log 2>&1 || echo "misc pass failed"
python3 scripts/summarize_prof.py $(ls -d $O/*/) > $O/summary.txt
tail -3 $O.trace.log
cat $O/summary.txt

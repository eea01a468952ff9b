#!/bin/bash
# rocprofv3 PMC passes of the two-pairs-per-lane fill kernel (run on the GPU box from the repo root).
# Usage: bash scripts/prof_lane.sh NAME [pairs]
set -e
NAME=${1:-lane}; PAIRS=${2:-2097152}
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --workspace-gib 128 --pairs $PAIRS ${BENCH_EXTRA:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.k.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O.s.log 2>&1 || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc_sq2 -- $B > $O.s2.log 2>&1 || echo "sq2 pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O.f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O.w.log 2>&1 || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds -- $B > $O.l.log 2>&1 || echo "lds pass failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $B > $O.t.log 2>&1 || echo "tcc pass failed"
python3 scripts/summarize_prof.py $(ls -d $O/*/) > $O/summary.txt
cat $O/summary.txt

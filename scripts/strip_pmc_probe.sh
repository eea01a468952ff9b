#!/bin/bash
# round 4: VALU instructions of the long-read fill (one pass of 768 pairs of 10 kb)
O=gpurun_out/${1:-r04_ilpmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 280 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/long_sq -- python3 scripts/long_read_bench.py 768 40 10000 0 > $O/long_sq.log 2>&1
python3 scripts/summarize_prof.py $O/long_sq/ | grep -A8 "strip_kernel" | head -12

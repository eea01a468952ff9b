#!/bin/bash
# round 4: how many of the headline kernel's VALU instructions are pass 2's (a -DMGL_CK_ABLATE=1 build skips pass 2's loop: wrong results, right count)
O=gpurun_out/${1:-r04_p2count}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --workspace-gib 8 --pairs 2097152"
timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/full -- $B > $O/full.log 2>&1
MGL_SW_LIB=build/variants/lib_ck_nop2.so timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/nop2 -- $B > $O/nop2.log 2>&1
python3 scripts/summarize_prof.py $O/full/ $O/nop2/ | grep -E "counters|lane_ck|INSTS" | cut -c1-150

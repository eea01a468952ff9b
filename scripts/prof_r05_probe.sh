#!/bin/bash
# round 5, first measurement session (run on the GPU box from the repo root): counters of the protein kernel (the review's item 5: "counters first"),
# the headline kernel's bytes / power experiment (item 7: pass 1 with and without the `mid` rows, pass 2 skipped in both -- variant builds),
# mixed read lengths from pinned host memory through the packed entry as it stands (item 4a's starting point).
NAME=${1:-r05_probe}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
run() { local label=$1; shift; timeout -k 10 280 rocprofv3 "$@" > $O/$label.log 2>&1 || echo "$label failed" | tee -a $O/failed.txt; }
PM="--output-format csv"
# ---- protein: one pass of 2 M alignments (sw_dp16_matrix_kernel + sw_traceback_kernel)
P="python3 scripts/protein_bench.py --steps 1 --check 0"
run prot_trace --kernel-trace --stats $PM -d $O/prot_trace -- $P
run prot_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES $PM -d $O/prot_sq -- $P
run prot_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/prot_lds -- $P
run prot_sq2 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES $PM -d $O/prot_sq2 -- $P
run prot_fetch --pmc FETCH_SIZE $PM -d $O/prot_fetch -- $P
run prot_write --pmc WRITE_SIZE $PM -d $O/prot_write -- $P
python3 scripts/summarize_prof.py $(ls -d $O/prot_*/) > $O/protein_summary.txt 2>&1
# ---- headline, item 7: pass 1 alone (MGL_CK_ABLATE=1) against pass 1 without the `mid` rows (MGL_CK_ABLATE=5): ms, clock, WRITE_SIZE
for v in ck_nop2 ck_nop2_nomid; do
  MGL_SW_LIB=build/variants/lib_$v.so REPS=5 FULL_ONLY=1 timeout -k 10 200 python3 scripts/lane_probe.py 10000000 > $O/$v.time.log 2>&1
  MGL_SW_LIB=build/variants/lib_$v.so timeout -k 10 200 python3 scripts/ck_clock_probe.py 10000000 > $O/$v.clock.log 2>&1
  B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --workspace-gib 8 --pairs 2097152"
  MGL_SW_LIB=build/variants/lib_$v.so run ${v}_write --pmc WRITE_SIZE $PM -d $O/${v}_write -- $B
  MGL_SW_LIB=build/variants/lib_$v.so run ${v}_fetch --pmc FETCH_SIZE $PM -d $O/${v}_fetch -- $B
done
python3 scripts/summarize_prof.py $(ls -d $O/ck_*/) > $O/headline_bytes_summary.txt 2>&1
# ---- mixed read lengths from pinned host memory, the packed entry as it stands
timeout -k 10 280 python3 scripts/mixed_host_probe.py 4000000 100 --json > $O/mixed_host.log 2>&1
MGL_SW_DEBUG_HOST_TIMING=1 timeout -k 10 280 python3 scripts/mixed_host_probe.py 4000000 100 > $O/mixed_host_timing.log 2>&1
cat $O/failed.txt 2>/dev/null
tail -3 $O/*.time.log $O/*.clock.log $O/mixed_host.log

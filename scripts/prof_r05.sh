#!/bin/bash
# round 5 evidence on the SHIPPED build (run on the GPU box from the repo root): kernel trace + PMC passes of the headline kernel at
# 256 x 150 and 1000 x 150, LDS-conflict and traffic passes of the long-read walk, the one-wave-per-pair kernel and PairHMM, and the
# kernel + copy timeline of the packed host entry.  Counter passes never share a run with a trace domain.
#   bash scripts/prof_r05.sh NAME
NAME=${1:-r05_prof}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
run() { # label, then the rocprofv3 arguments up to --, then the program
  local label=$1; shift
  timeout -k 10 280 rocprofv3 "$@" > $O/$label.log 2>&1 || echo "$label failed" | tee -a $O/failed.txt
}
PM="--output-format csv"
# ---- headline kernel, one dispatch of 2 097 152 pairs (256 x 150) and of 524 288 pairs (1000 x 150)
for shape in "256 2097152 8" "1000 524288 24"; do
  set -- $shape; TL=$1; PAIRS=$2; WS=$3
  B="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --no-extra --workspace-gib $WS --pairs $PAIRS --tl $TL"
  run lane${TL}_trace --kernel-trace --stats $PM -d $O/lane${TL}_trace -- $B
  run lane${TL}_sq  --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES $PM -d $O/lane${TL}_sq -- $B
  run lane${TL}_sq2 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS $PM -d $O/lane${TL}_sq2 -- $B
  run lane${TL}_fetch --pmc FETCH_SIZE $PM -d $O/lane${TL}_fetch -- $B
  run lane${TL}_write --pmc WRITE_SIZE $PM -d $O/lane${TL}_write -- $B
  run lane${TL}_tcc --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum $PM -d $O/lane${TL}_tcc -- $B
done
# ---- long reads: 768 pairs of 10 kb (one round of the chip), one pass: fill + walk
L="python3 scripts/long_read_bench.py 768 40 10000 0"
run long_trace --kernel-trace --stats $PM -d $O/long_trace -- $L
run long_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES $PM -d $O/long_sq -- $L
run long_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/long_lds -- $L
run long_fetch --pmc FETCH_SIZE $PM -d $O/long_fetch -- $L
run long_write --pmc WRITE_SIZE $PM -d $O/long_write -- $L
# ---- one wave per pair (sw_small_kernel): 4 096 pairs of 256 x 150 per launch
S="python3 scripts/small_kernel_probe.py 4096"
run small_trace --kernel-trace --stats $PM -d $O/small_trace -- $S
run small_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES $PM -d $O/small_sq -- $S
run small_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/small_lds -- $S
# ---- PairHMM float kernel, 1.6 M pairs of 150 x ~300
P="python3 scripts/pairhmm_bench.py --no-cpu --steps 1"
run pairhmm_trace --kernel-trace --stats $PM -d $O/pairhmm_trace -- $P
run pairhmm_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES $PM -d $O/pairhmm_sq -- $P
run pairhmm_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/pairhmm_lds -- $P
# ---- protein (configs[4] shape): one pass of 2 M alignments in the shared-target layout: sw_dp16_lane_matrix_kernel (1.92 M pairs) + sw_dp16_matrix_kernel + sw_traceback_kernel (80 000)
Q="python3 scripts/protein_bench.py --steps 1 --check 0"
run prot_trace --kernel-trace --stats $PM -d $O/prot_trace -- $Q
run prot_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES $PM -d $O/prot_sq -- $Q
run prot_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS $PM -d $O/prot_lds -- $Q
run prot_fetch --pmc FETCH_SIZE $PM -d $O/prot_fetch -- $Q
run prot_write --pmc WRITE_SIZE $PM -d $O/prot_write -- $Q
# ---- the host entries: kernels and copies on one timeline (trace domains only, no counters): the packed entry, the ASCII entry's direct form, mixed read lengths
run ascii_timeline --kernel-trace --memory-copy-trace $PM -d $O/ascii_timeline -- python3 scripts/ascii_direct_probe.py 10000000 262144
run mixed_timeline --kernel-trace --memory-copy-trace $PM -d $O/mixed_timeline -- python3 scripts/mixed_host_probe.py 4000000 100
# ---- the packed host entry: kernels and copies on one timeline (trace domains only, no counters)
run host_timeline --kernel-trace --memory-copy-trace $PM -d $O/host_timeline -- python3 scripts/host_packed_probe.py 10000000 2
python3 scripts/summarize_prof.py $(ls -d $O/*/ | grep -v _timeline) > $O/summary.txt 2>&1
python3 scripts/host_timeline.py $O/ascii_timeline > $O/ascii_timeline.txt 2>&1
python3 scripts/host_timeline.py $O/mixed_timeline > $O/mixed_timeline.txt 2>&1
python3 scripts/host_timeline.py $O/host_timeline > $O/host_timeline.txt 2>&1
tail -40 $O/host_timeline.txt
cat $O/failed.txt 2>/dev/null
echo "commit $(cat $R/.commit_id 2>/dev/null)" >> $O/summary.txt

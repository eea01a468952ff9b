#!/bin/bash
# rocprofv3 kernel traces of the secondary workloads (GPU box, repo root): bash scripts/prof_secondary.sh NAME
set -e
NAME=${1:-prof_sec}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
echo "long reads"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/long -- python3 scripts/long_read_bench.py 256 32 10000 1 > $O.long.log 2>&1
echo "pairhmm";    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pairhmm -- python3 scripts/pairhmm_bench.py --no-cpu > $O.pairhmm.log 2>&1
echo "protein";    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/protein -- python3 scripts/protein_bench.py --check 0 > $O.protein.log 2>&1
python3 scripts/summarize_prof.py $O/long $O/pairhmm $O/protein > $O/summary.txt
cat $O/summary.txt
# one PMC pass each (SQ counters): VALU instruction counts of the secondary kernels
echo "pmc long reads"; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_long -- python3 scripts/long_read_bench.py 256 32 10000 1 > $O.pmc_long.log 2>&1
echo "pmc pairhmm";    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_pairhmm -- python3 scripts/pairhmm_bench.py --no-cpu --steps 1 > $O.pmc_pairhmm.log 2>&1
python3 scripts/summarize_prof.py $O/pmc_long $O/pmc_pairhmm > $O/summary_pmc.txt
cat $O/summary_pmc.txt

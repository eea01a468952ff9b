#!/bin/bash
# every secondary measurement quoted in DESIGN.md, one after the other (GPU box, repo root); logs under gpurun_out/sec/
set -e
O=gpurun_out/sec; mkdir -p $O
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > $O/$name.log 2>&1; tail -3 $O/$name.log; }
run extra python scripts/extra_measurements.py
run protein python scripts/protein_bench.py
run protein_score python scripts/protein_bench.py --score-only
export MGL_PROTEIN_INT32=1; run protein_int32 python scripts/protein_bench.py; unset MGL_PROTEIN_INT32
run grouped python scripts/grouped_bench.py
run long2048 python scripts/long_read_bench.py 2048 220 --seconds 20
run big_fuzz python scripts/big_fuzz.py

#!/bin/bash
# Timeline of the host-buffer entry (mgl_sw_align_batch, PCIe inclusive): kernels and memory copies of the same run,
# to see how far the copies hide behind the kernels.  bash scripts/prof_host_entry.sh NAME [extra env assignments ...]
set -e
NAME=${1:-r02_host}; shift || true
R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 scripts/host_entry_probe.py 10000000 0 200 > $O.log 2>&1 || echo "trace failed"
tail -4 $O.log
python3 scripts/host_timeline.py $O/trace > $O/timeline.txt
cat $O/timeline.txt

#!/bin/bash
# The GPU test suite with a post-mortem: when the process dies of a GPU memory fault the runtime leaves a GPU core (gpucore.PID) in the
# working directory; rocgdb reads it and says which kernel's waves were where.  bash scripts/gpu_suite.sh NAME [pytest arguments]
NAME=${1:-suite}; shift
O=gpurun_out/$NAME; mkdir -p $O
rm -f gpucore.*
python -m pytest tests -x -q -m gpu "$@" > $O/pytest.txt 2>&1
rc=$?
if [ $rc -ne 0 ]; then
  ls -la gpucore.* > $O/cores.txt 2>&1
  for c in gpucore.*; do
    [ -f "$c" ] || continue
    timeout -k 10 240 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info queues" -ex "info dispatches" -ex "info threads" -ex "thread apply all bt 6" \
      "$(command -v python3)" "$c" > $O/rocgdb_$c.txt 2>&1
  done
fi
tail -3 $O/pytest.txt
exit $rc

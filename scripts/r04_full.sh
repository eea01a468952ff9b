#!/bin/bash
# round 4: the whole GPU suite, the bench line, and the timeline of the packed host entry (kernel + copy trace; no counters)
set -o pipefail
O=gpurun_out/${1:-r04g}; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/gpu_tests.log | tee -a $O/summary.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - $O <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
    p = d.get("pcie_inclusive", {})
    print("value", d["value"], "ms", d["ms_per_step"], "pcie", p.get("gcups"), (p.get("registered") or {}).get("gcups"), (p.get("packed_2bit") or {}).get("gcups"), "tl1000", d.get("tl1000", {}).get("gcups"))
    for k, v in d.get("secondary", {}).items():
        print("  ", k[:60], {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)})
except Exception as e:
    print("bench parse failed", e)
PY

#!/bin/bash
# round 4, first GPU pass: parity of the persistent grid + column pairs, then the A/B timing against the single-column build
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane or 2bit or grouped or mixed or regroup or sorted" > $O/parity_lane.log 2>&1; echo "parity_lane rc=$?" | tee -a $O/summary.txt
tail -3 $O/parity_lane.log | tee -a $O/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
tail -3 $O/fullsize.log | tee -a $O/summary.txt
export LANE_MODE=0 FULL_ONLY=1
for n in 10000000 1250000 2097152; do
  for rep in 1 2; do
    echo -n "pairs $n regular: " | tee -a $O/summary.txt; timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
    echo -n "pairs $n nopair : " | tee -a $O/summary.txt; MGL_SW_LIB=$PWD/build/variants/lib_nopair.so timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
  done
done
echo -n "tl1000 regular: " | tee -a $O/summary.txt; WS_GIB=24 timeout -k 10 200 python scripts/lane_probe.py 2560000 1000 150 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
echo -n "tl1000 nopair : " | tee -a $O/summary.txt; WS_GIB=24 MGL_SW_LIB=$PWD/build/variants/lib_nopair.so timeout -k 10 200 python scripts/lane_probe.py 2560000 1000 150 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a gpurun_out/r04a/summary.txt
import json
try:
    d = json.loads(open("gpurun_out/r04a/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "pcie", {k: (v.get("gcups") if isinstance(v, dict) else v) for k, v in d.get("pcie_inclusive", {}).items() if k in ("gcups", "registered", "packed_2bit")}, "tl1000", d.get("tl1000", {}).get("gcups"))
except Exception as e:
    print("bench parse failed", e)
PY

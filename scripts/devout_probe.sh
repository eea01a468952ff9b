#!/bin/bash
# round 4: device-resident 2-bit inputs against ASCII (the packed host entry's kernel stages 2-bit inputs)
set -o pipefail
O=gpurun_out/${1:-r04_devout}; mkdir -p $O
for inp in ascii 2bit ascii 2bit; do
  echo "== device resident, --input $inp" | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-extra --input $inp 2>/dev/null | grep '^{"metric"' | python3 -c "import sys, json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/summary.txt
done

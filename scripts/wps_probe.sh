export FULL_ONLY=1 LANE_MODE=0
for n in 1250000 2500000 5000000 786432 393216 3000000; do for w in 3 2; do echo -n "n=$n wps=$w: "; MGL_SW_LANE_CK_WPS=$w timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "per call = [0-9]* GCUPS"; done; done

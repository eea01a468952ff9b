# host-buffer entry, 10 M uniform pairs, PCIe inclusive: chunk count x result-copy thread x two fill streams
export MGL_SW_HOST_TIMING=1
for c in ${CHUNKS:-8 12}; do
  for t in ${THREADS:-1 0}; do
    for d in ${DUAL:-1 0}; do
      echo "== chunks $c out_thread $t dual $d"
      MGL_SW_HOST_CHUNKS=$c MGL_SW_OUT_THREAD=$t MGL_SW_LANE_DUAL=$d timeout -k 10 300 python scripts/host_entry_probe.py 10000000 0 200 2>&1 | grep -v amdgpu.ids | tail -2
    done
  done
done

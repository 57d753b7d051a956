mkdir -p gpurun_out
for L in panini hammer; do
  RES_DEBUG=1 timeout 120 python tools/resident_traffic.py $L 2000 3840 2160 8192 > gpurun_out/r05_busy_$L.log 2>&1
  mv gpurun_out/res_wg_stats.txt gpurun_out/r05_wg_stats_$L.txt
done
tail -2 gpurun_out/r05_busy_panini.log

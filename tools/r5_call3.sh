mkdir -p gpurun_out
for F in 8192 73728 139264 270336 401408; do
  RES_DEBUG=1 timeout 120 python tools/resident_traffic.py panini 2000 3840 2160 $F > gpurun_out/r05_busy_panini_$F.log 2>&1
  mv gpurun_out/res_wg_stats.txt gpurun_out/r05_wg_stats_panini_$F.txt
  grep "RESIDENT panini" gpurun_out/r05_busy_panini_$F.log
done

mkdir -p gpurun_out; rm -f gpurun_out/bal3_probe.log
timeout 900 python -m pytest tests/test_apply_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/bal3_tests.log
for l in hammer quincuncial panini; do
  BK_RING=64 BK_WGS=16 BK_REPS=200 BK_ABLATE=64,0,64,0 timeout 300 python tools/apply_probe.py $l 3840 2160 16 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats.*//" >> gpurun_out/bal3_probe.log
done
BK_RING=64 BK_WGS=16 BK_REPS=200 BK_ABLATE=64,0,64,0 timeout 300 python tools/apply_probe.py stereographic 1920 1080 16 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats.*//" >> gpurun_out/bal3_probe.log
cat gpurun_out/bal3_tests.log gpurun_out/bal3_probe.log

mkdir -p gpurun_out; rm -f gpurun_out/bal6_probe.log
BK_RING=64 BK_REPS=10 BK_WGS=16,128 timeout 300 python tools/apply_probe.py hammer 7680 4320 64 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats: / /" >> gpurun_out/bal6_probe.log
BK_RING=64 BK_REPS=20 BK_WGS=16 timeout 300 python tools/apply_probe.py panini 7680 4320 16 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats: / /" >> gpurun_out/bal6_probe.log
for l in hammer quincuncial stereographic; do BK_RING=64 BK_REPS=100 BK_WGS=16 timeout 300 python tools/apply_probe.py $l 3840 2160 16 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats: / /" >> gpurun_out/bal6_probe.log; done
BK_RING=64 BK_REPS=100 BK_WGS=16 timeout 300 python tools/apply_probe.py hammer 3840 2160 64 2 2>&1 | grep -E "ablation|frames/launch" | sed "s/; tile stats: / /" >> gpurun_out/bal6_probe.log
cat gpurun_out/bal6_probe.log

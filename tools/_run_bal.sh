mkdir -p gpurun_out; rm -f gpurun_out/r2e.log
run() { echo "== $*" >> gpurun_out/r2e.log; "$@" 2>&1 | grep -E "frames/launch" >> gpurun_out/r2e.log; }
export BK_RING=64 BK_REPS=300 BK_WGS=16,17,16,17
run python tools/apply_probe.py panini 3840 2160 1 2
run python tools/apply_probe.py quincuncial 3840 2160 1 2
run python tools/apply_probe.py hammer 3840 2160 1 2
BK_GLOBE=trism run python tools/apply_probe.py panini 3840 2160 1 2
cat gpurun_out/r2e.log

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_build_gpu.py tests/test_build_campaign_gpu.py tests/test_host_path_gpu.py -q -x -m gpu 2>&1 | tail -3
python tools/build_probe.py --lenses winkel2,eckert1,polyconic,panini 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $GRAFT_REPO_ROOT/tools/build_probe.py --lenses winkel2,polyconic --reps 2 > /dev/null 2>&1
db=$(find /tmp/kt1 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py "$db" 2>&1 | head -12

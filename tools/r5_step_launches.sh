# every launch of one config-5 step (kernel trace of tools/prof2d.py, tools/trace_gaps.py): bash tools/r5_step_launches.sh [out file]
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ksl
rocprofv3 --kernel-trace -d /tmp/ksl -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/ksl.log 2>&1
cd /root/repo; mkdir -p gpurun_out; python3 tools/trace_gaps.py $(find /tmp/ksl -name "*.db" | head -1) stem7 | tee ${1:-gpurun_out/step_launches.txt}

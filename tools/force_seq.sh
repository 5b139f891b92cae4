# per-launch sequence of one ForceUnet design-gradient call (rocprofv3 kernel trace of tools/bench_force.py)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kf
rocprofv3 --kernel-trace -d /tmp/kf -o cf -- python3 /root/repo/tools/bench_force.py 64 2 > /tmp/kf.log 2>&1
cd /root/repo; mkdir -p gpurun_out/r2x
python3 tools/rocprof_sequence.py $(find /tmp/kf -name "*.db" | head -1) "fu_conv_kernel<7, 4, 0>" > gpurun_out/r2x/force_seq.txt
grep -c . gpurun_out/r2x/force_seq.txt

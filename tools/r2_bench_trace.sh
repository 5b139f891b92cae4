# rocprofv3 kernel trace of the bench command itself (config 2 and config 5), summarised per kernel
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kb2; rocprofv3 --kernel-trace -d /tmp/kb2 -o b2 -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kb2.log 2>&1
# (config 5: rocprofv3 itself segfaults while tracing bench.py --workload cfg5 on this image; tools/gpu_round_measure.sh traces tools/prof2d.py instead)
cd /root/repo; mkdir -p gpurun_out/meas
(echo "# rocprofv3 --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline   (under the profiler; bench line below)"; tail -1 /tmp/kb2.log | cut -c1-400; python3 tools/rocprof_summary.py $(find /tmp/kb2 -name "*.db" | head -1)) > gpurun_out/meas/bench_trace_cfg2.txt
head -12 gpurun_out/meas/bench_trace_cfg2.txt | cut -c1-150; head -8 gpurun_out/meas/bench_trace_cfg5.txt | cut -c1-150

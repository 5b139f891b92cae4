# rocprofv3 kernel trace of the 1-D sampling loop (config 2): per-kernel stats -> gpurun_out/r2p/kstats_cfg2.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/k2; rocprofv3 --kernel-trace -d /tmp/k2 -o c2 -- python3 /root/repo/tools/prof1d.py 256 50 > /tmp/k2.log 2>&1
cd /root/repo; mkdir -p gpurun_out/r2p
python3 tools/rocprof_summary.py $(find /tmp/k2 -name "*.db" | head -1) gpurun_out/r2p/kstats_cfg2.txt | cut -c1-150
tail -2 /tmp/k2.log

cd /root/repo
for v in 1 0; do
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/k2
CINDM_L2_PREFETCH=$v rocprofv3 --kernel-trace -d /tmp/k2 -o c2 -- python3 /root/repo/tools/prof1d.py 256 50 > /tmp/k2.log 2>&1
cd /root/repo; mkdir -p gpurun_out/r2p
echo "== prefetch=$v"; python3 tools/rocprof_summary.py $(find /tmp/k2 -name "*.db" | head -1) gpurun_out/r2p/kstats_pf$v.txt | cut -c1-130 | grep -E "ups_tail|dconv_kernel<6, 2, 0, false>|dconv_kernel<3, 4, 0, false>|level1"
done

cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 3 4; do
  export CINDM_DBG2=$d
  rocprofv3 --kernel-trace -d /tmp/r$d -o c5 -- python3 /root/repo/tools/prof2d.py 64 2 6 > /dev/null 2>&1
  echo DBG2=$d; python3 /root/repo/tools/rocprof_summary.py $(find /tmp/r$d -name "*.db" | head -1) | grep "conv2d_h3_kernel<0"
done

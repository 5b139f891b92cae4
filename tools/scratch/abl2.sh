cd /tmp; export TMPDIR=/tmp
for d in 4 5; do
  export CINDM_DBG3=$d
  rocprofv3 --kernel-trace -d /tmp/q$d -o c5 -- python3 /root/repo/tools/prof2d.py 64 2 6 > /dev/null 2>&1
  echo DBG3=$d; python3 /root/repo/tools/rocprof_summary.py $(find /tmp/q$d -name "*.db" | head -1) | grep "la2d_context"
done

cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 3; do
  export CINDM_DBG3=$d
  rocprofv3 --kernel-trace -d /tmp/p$d -o site -- python3 /root/repo/tools/prof1d.py 256 30 > /dev/null 2>&1
  echo DBG3=$d; python3 /root/repo/tools/rocprof_summary.py $(find /tmp/p$d -name "*.db" | head -1) | grep site
done

# LA kernels at C = 64: workgroups per image.  bash tools/r5_exp_la.sh <option> <kernel filter> <values...>
opt=$1; filt=$2; shift 2
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/kla
  rocprofv3 --kernel-trace -d /tmp/kla -o t -- python3 /root/repo/tools/ab2d.py $opt $v $v 5 > /tmp/kla.log 2>&1
  (cd /root/repo; echo "$opt=$v"; python3 tools/rocprof_summary.py $(find /tmp/kla -name "*.db" | head -1) | grep "$filt")
done

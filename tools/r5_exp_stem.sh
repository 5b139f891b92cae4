# kernel durations under both values of one 2-D option in one process: bash tools/r5_exp_stem.sh <option> [filter]
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kst
rocprofv3 --kernel-trace -d /tmp/kst -o t -- python3 /root/repo/tools/ab2d.py ${1:-stem_dense} ${3:-0} ${4:-1} 10 > /tmp/kst.log 2>&1
cd /root/repo; python3 tools/rocprof_summary.py $(find /tmp/kst -name "*.db" | head -1) | grep -i "${2:-stem}"

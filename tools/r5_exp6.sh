cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e9; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py -x -q -m gpu > $out/test2d.txt 2>&1; tail -3 $out/test2d.txt
bash tools/r5_ab_trees.sh r5e9 2>&1 | tail -8
grep -v "1\.0[0-2][0-9]$\|0\.9[89][0-9]$" $out/ratio.txt | head -40

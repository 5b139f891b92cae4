cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e14; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py -x -q -m gpu > $out/test2d.txt 2>&1; tail -3 $out/test2d.txt
timeout 900 python -m pytest tests/test_gpu_force.py -x -q -m gpu -k "golden or fp32 or other_image or fused_linear" > $out/testforce.txt 2>&1; tail -3 $out/testforce.txt
bash tools/r5_ab_trees.sh r5e14 2>&1 | tail -8
grep "la2d" $out/ratio.txt
for t in _before .; do (cd /root/repo/$t && timeout 600 python tools/bench_force.py 64 2 5 2>/dev/null | grep "design gradient" | sed "s#^#$t #"); done

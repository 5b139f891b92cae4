cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e11; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity.py -x -q -m gpu > $out/tests.txt 2>&1; tail -3 $out/tests.txt
timeout 900 python -m pytest tests/test_gpu_force.py -x -q -m gpu -k "golden or fp32 or stress" > $out/testforce.txt 2>&1; tail -3 $out/testforce.txt
bash tools/r5_ab_trees.sh r5e11 2>&1 | tail -8

cd /root/repo; timeout 1200 python -m pytest tests/test_gpu_range.py tests/test_gpu_parity.py tests/test_gpu_parity_2d.py -x -q 2>&1 | tail -8

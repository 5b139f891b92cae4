cd /root/repo; mkdir -p gpurun_out/r2t; rm -f gpurun_out/r2t/*
timeout 900 python -m pytest tests/test_gpu_range.py -m gpu -q > gpurun_out/r2t/pytest_new.txt 2>&1; tail -30 gpurun_out/r2t/pytest_new.txt | cut -c1-200

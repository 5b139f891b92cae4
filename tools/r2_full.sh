cd /root/repo; mkdir -p gpurun_out/r2f; rm -f gpurun_out/r2f/*
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest.txt 2>&1; tail -4 gpurun_out/r2f/pytest.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-400

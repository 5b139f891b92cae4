cd /root/repo; mkdir -p gpurun_out/r2x; rm -f gpurun_out/r2x/pytest.txt
timeout 900 python -m pytest tests/test_gpu_force.py -m gpu -q -x > gpurun_out/r2x/pytest.txt 2>&1; tail -12 gpurun_out/r2x/pytest.txt | cut -c1-220
python3 tools/bench_force.py 64 2 2>&1 | grep -v amdgpu.ids

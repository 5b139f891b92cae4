cd /root/repo; mkdir -p gpurun_out/r2g; rm -f gpurun_out/r2g/*
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2g/pytest.txt 2>&1; tail -3 gpurun_out/r2g/pytest.txt
for v in 1 0 1 0; do CINDM_L2_PREFETCH=$v python3 tools/prof1d.py 256 300 2>&1 | grep -v amdgpu.ids | sed "s/^/pf=$v /"; done

cd /root/repo; mkdir -p gpurun_out/r2i; rm -f gpurun_out/r2i/*
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet or chain_cfg1" > gpurun_out/r2i/pytest.txt 2>&1; tail -3 gpurun_out/r2i/pytest.txt
for i in 1 2; do python3 tools/prof1d.py 256 300 2>&1 | grep -v amdgpu.ids; done
python3 tools/gpu_layers.py 256 2>&1 | grep "conv5" | head -20

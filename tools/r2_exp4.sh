cd /root/repo; mkdir -p gpurun_out/r2d; rm -f gpurun_out/r2d/*
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet" > gpurun_out/r2d/pytest_unet.txt 2>&1; tail -3 gpurun_out/r2d/pytest_unet.txt
python3 tools/gpu_layers.py 256 > gpurun_out/r2d/layers.txt 2>&1; tail -30 gpurun_out/r2d/layers.txt
python3 tools/prof1d.py 256 200 > gpurun_out/r2d/step.txt 2>&1; cat gpurun_out/r2d/step.txt

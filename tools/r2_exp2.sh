cd /root/repo; mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet" > gpurun_out/r2b/pytest_unet.txt 2>&1
tail -15 gpurun_out/r2b/pytest_unet.txt
python3 tools/gpu_layers.py 256 > gpurun_out/r2b/layers.txt 2>&1
CINDM_DCONV_PAIR=0 python3 tools/gpu_layers.py 256 > gpurun_out/r2b/layers_nopair.txt 2>&1
python3 tools/prof1d.py 256 200 > gpurun_out/r2b/step.txt 2>&1; cat gpurun_out/r2b/step.txt

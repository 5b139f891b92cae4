# round-2 experiment 1: weight-stream microbenchmark, main-loop ablations of the k=5 kernel, GPU suite
cd /root/repo; mkdir -p gpurun_out/r2a
./tools/micro/bstream.bin > gpurun_out/r2a/bstream.txt 2>&1
for d in 0 21 22 23 24 25; do echo "== DBG=$d" >> gpurun_out/r2a/layers.txt; CINDM_DBG=$d python3 tools/gpu_layers.py 256 >> gpurun_out/r2a/layers.txt 2>&1; done
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.txt 2>&1
tail -3 gpurun_out/r2a/pytest.txt
cat gpurun_out/r2a/bstream.txt

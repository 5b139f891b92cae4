cd /root/repo; mkdir -p gpurun_out/r2c; rm -f gpurun_out/r2c/variants.txt
for v in "24 8 1 33" "24 8 1 10" "24 8 1 1" "24 8 0 9" "24 8 0 33" "24 8 1 2 dconv=0"; do
  echo "== $v" >> gpurun_out/r2c/variants.txt
  python3 tests/debug/variants.py $v >> gpurun_out/r2c/variants.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r2c/variants.txt | tail -40
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.txt 2>&1; tail -5 gpurun_out/r2c/pytest.txt
python3 tools/gpu_layers.py 256 > gpurun_out/r2c/layers.txt 2>&1
python3 tools/prof1d.py 256 200 > gpurun_out/r2c/step.txt 2>&1; cat gpurun_out/r2c/step.txt

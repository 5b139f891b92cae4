cd /root/repo; mkdir -p gpurun_out/r2h; rm -f gpurun_out/r2h/*
for v in "24 8 1 33" "24 8 1 10" "24 8 1 1" "24 8 1 4" "24 8 1 33 attn_head=0"; do
  echo "== $v" >> gpurun_out/r2h/variants.txt
  timeout 120 python3 tests/debug/variants.py $v >> gpurun_out/r2h/variants.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r2h/variants.txt | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2h/pytest.txt 2>&1; tail -3 gpurun_out/r2h/pytest.txt
for v in 1 0; do CINDM_ATTN_HEAD=$v timeout 120 python3 tools/prof1d.py 256 300 2>&1 | grep -v amdgpu.ids | sed "s/^/attn_head=$v /"; done
timeout 120 python3 tools/gpu_layers.py 256 2>&1 | grep "attention_and" 

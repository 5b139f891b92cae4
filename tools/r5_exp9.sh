cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e13; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py -x -q -m gpu > $out/test2d.txt 2>&1; tail -3 $out/test2d.txt
timeout 300 python tools/ab2d.py ws_store3 0 1 20 > $out/ab_store3.txt 2>&1; grep -v amdgpu $out/ab_store3.txt
CINDM_LIB_VARIANT=prof timeout 300 python tools/ws_prof.py 128 3 2 0 > $out/ws_prof_store3.txt 2>&1; grep -v "^/opt" $out/ws_prof_store3.txt
timeout 1200 python -m pytest tests/test_gpu_force.py -x -q -m gpu -k "golden or fp32 or stress or other_image or full_batch" > $out/testforce.txt 2>&1; tail -3 $out/testforce.txt
timeout 900 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -k "unet2d" > $out/testpaths.txt 2>&1; tail -3 $out/testpaths.txt
timeout 600 python tools/bench_force.py 64 2 5 > $out/force.txt 2>&1; tail -3 $out/force.txt

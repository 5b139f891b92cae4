# Round 6, experiment 7: conv2d_ws_kernel's tile stores written through (dbg2 = 31) -- cfg5 step time
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python tools/ab2d.py dbg2 0 31 20 > $out/ab2d_wt.txt 2>&1; cat $out/ab2d_wt.txt | grep ms/step
timeout 900 python -m pytest tests/test_gpu_range.py -x -q -k "callers_first or beyond" 2>&1 | tail -3

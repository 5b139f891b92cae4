cd /root/repo; mkdir -p gpurun_out
(timeout 2700 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -45; python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3) | tee gpurun_out/r06_gpu_tests.txt

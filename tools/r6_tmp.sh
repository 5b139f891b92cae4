cd /root/repo; mkdir -p gpurun_out/r6
CINDM_LIB_VARIANT=prof PHASE_OPTS=l2_prefetch=0 timeout 300 python tools/phase_table.py cfg2 40 > gpurun_out/r6/phase_nopf.txt 2>/dev/null
CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py cfg2 40 > gpurun_out/r6/phase_pf.txt 2>/dev/null
tail -1 gpurun_out/r6/phase_nopf.txt gpurun_out/r6/phase_pf.txt

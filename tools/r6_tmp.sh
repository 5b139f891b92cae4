cd /root/repo; mkdir -p gpurun_out/r6
CINDM_LIB_VARIANT=prof PHASE_RAW=ups_last timeout 300 python tools/phase_table.py cfg2 40 2>/dev/null | grep "stamps relative"
CINDM_LIB_VARIANT=prof PHASE_RAW=ups_last PHASE_OPTS=fuse_update=0 timeout 300 python tools/phase_table.py cfg2 40 2>/dev/null | grep "stamps relative"

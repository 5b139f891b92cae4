# Phase ablation of the k=5 convolution launches at batch 4 (pure latency: 2-16 workgroups per launch).
# CINDM_DBG: 10 return at entry, 11 after the prologue (first stage staged), 9 no epilogue, 8 no statistics, 7 no main loop
for d in 0 10 11 7 9 8; do echo DBG=$d; CINDM_DBG=$d python3 tools/gpu_layers.py 4 2>&1 | sed -n '3,5p;14,16p;21,23p'; done

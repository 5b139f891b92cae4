# rocprofv3 kernel trace of the 2-D sampling loop (config 5): per-launch sequence + per-kernel stats + phase ablations
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/k5; rocprofv3 --kernel-trace -d /tmp/k5 -o c5 -- python3 /root/repo/tools/prof2d.py 64 2 6 > /tmp/k5.log 2>&1
cd /root/repo; mkdir -p gpurun_out/r2g
DB=$(find /tmp/k5 -name "*.db" | head -1)
python3 tools/rocprof_summary.py $DB gpurun_out/r2g/kstats_cfg5.txt | cut -c1-150 | head -30
python3 tools/rocprof_sequence.py $DB > gpurun_out/r2g/seq_cfg5.txt
tail -2 /tmp/k5.log
for d in 2 3 4 5; do
python3 - <<PY
import os, sys, torch
sys.path.insert(0, "/root/repo"); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
x = torch.randn((128, 4096, 24), device=dev); x[:, :, 21:] = 0
m.profile(x, 500)
base = m.profile(x, 500)
m.set_option("dbg2", $d)
m.profile(x, 500)
r = m.profile(x, 500)
print("dbg2=$d conv3x3 %.0f us (full %.0f us)" % (r["conv3x3"][1] * 1e3, base["conv3x3"][1] * 1e3))
PY
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2g/phases_cfg5.txt

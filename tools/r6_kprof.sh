# Round 6, late: staging of phase A's k-steps CINDM_STAGE_LEAD k-steps ahead of their multiplication (kernels_dconv.h; library variants
# exp0 = 0 = as before, default = 1, exp2 = 2, exp3 = 3).  Alternating processes on one box, results compared bitwise, then the clocks
# inside phase A's K loop (variant kprof).
cd /root/repo; mkdir -p gpurun_out/r6
cat > gpurun_out/r6/chain.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
out = {}
for B in (256, 37, 768, 5):
    out[B] = d.sample(batch_size=B, seed=1, t_stop=960, n_composed=0, compose_n_bodies=2).cpu()
torch.save(out, sys.argv[1])
PY
python gpurun_out/r6/chain.py /tmp/new.pt; CINDM_LIB_VARIANT=exp0 python gpurun_out/r6/chain.py /tmp/old.pt; CINDM_LIB_VARIANT=exp3 python gpurun_out/r6/chain.py /tmp/e3.pt
python -c "
import torch
a, b, c = torch.load('/tmp/new.pt'), torch.load('/tmp/old.pt'), torch.load('/tmp/e3.pt')
print({B: bool(torch.equal(a[B], b[B]) and torch.equal(c[B], b[B])) for B in a}, 'lead 1 and lead 3 bitwise equal to lead 0')"
for r in 1 2 3; do
  for v in exp0 "" exp2 exp3; do echo -n "lead variant '$v': "; CINDM_LIB_VARIANT=$v python tools/ab1d.py tune 0 0 600 cfg2 2>/dev/null | grep us/step | tail -1; done
done
for v in exp0 "" exp2 exp3; do echo -n "cfg3 lead variant '$v': "; CINDM_LIB_VARIANT=$v python tools/ab1d.py tune 0 0 300 cfg3 2>/dev/null | grep us/step | tail -1; done
CINDM_LIB_VARIANT=kprof timeout 300 python tools/phase_table.py cfg2 40 > gpurun_out/r6/kprof_cfg2_lead1.txt 2> gpurun_out/r6/kprof.err; grep -A12 "mid_block1\|downs.3.1" gpurun_out/r6/kprof_cfg2_lead1.txt

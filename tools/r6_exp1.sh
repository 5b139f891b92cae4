# Round 6, experiment 1 (dconv2_kernel, option "tune"): bit 1 = poll / y0 requests before phase B's first weights; bit 2 = in-launch
# L2 warm-up of the launch's own fragments; bit 3 = phase B's share behind pair exchange A.  Same process, same box.
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python - > $out/tune_bitwise.txt 2>&1 <<'PY'
import torch, cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
ref = None
for t in (0, 2, 4, 12, 6, 14, 1):
    m.set_option("tune", t)
    x = d.sample(batch_size=256, seed=1, t_stop=960, n_composed=0, compose_n_bodies=2)
    torch.cuda.synchronize()
    if ref is None: ref = x.clone()
    print("tune", t, "bitwise equal to tune 0:", bool(torch.equal(x, ref)), "finite", bool(torch.isfinite(x).all()), flush=True)
PY
cat $out/tune_bitwise.txt
for t in 2 4 12 6 14; do python tools/ab1d.py tune 0 $t 600 cfg2 | grep us/step; done > $out/ab_tune.txt 2>&1
cat $out/ab_tune.txt
python tools/ab1d.py l2_prefetch 1 0 600 cfg2 | grep us/step > $out/ab_l2pf.txt 2>&1; cat $out/ab_l2pf.txt
CINDM_LIB_VARIANT=prof PHASE_SPLIT_NT=1 timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_cfg2_tune0.txt 2> $out/phase0.err
CINDM_LIB_VARIANT=prof PHASE_SPLIT_NT=1 PHASE_OPTS=tune=4 timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_cfg2_tune4.txt 2> $out/phase4.err
CINDM_LIB_VARIANT=prof PHASE_SPLIT_NT=1 PHASE_OPTS=tune=14 timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_cfg2_tune14.txt 2> $out/phase14.err
head -3 $out/phase_cfg2_tune0.txt; tail -1 $out/phase_cfg2_tune0.txt; tail -1 $out/phase_cfg2_tune4.txt; tail -1 $out/phase_cfg2_tune14.txt

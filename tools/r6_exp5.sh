# Round 6, experiment 5: the deep phase-B weight ring + one publishing wave (default) against round 5's structure (tune bit 1 = 2)
# and round 5's warm-up placement (tune bit 0 = 1).  3 = both = the round-5 kernel behaviour.
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python - > $out/tune_bitwise5.txt 2>&1 <<'PY'
import torch, cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
ref = None
for t in (3, 0, 1, 2):
    m.set_option("tune", t)
    x = d.sample(batch_size=256, seed=1, t_stop=900, n_composed=0, compose_n_bodies=2)
    torch.cuda.synchronize()
    if ref is None: ref = x.clone()
    print("tune", t, "bitwise equal to tune 3:", bool(torch.equal(x, ref)), "finite", bool(torch.isfinite(x).all()), "recovered", m.exchange_timeouts_recovered if hasattr(m, "exchange_timeouts_recovered") else None, flush=True)
PY
cat $out/tune_bitwise5.txt
for t in 0 2 1; do python tools/ab1d.py tune 3 $t 600 cfg2 | grep us/step; done > $out/ab_tune5.txt 2>&1
cat $out/ab_tune5.txt
python tools/ab1d.py tune 3 0 300 cfg3 | grep us/step > $out/ab_tune5_cfg3.txt 2>&1; cat $out/ab_tune5_cfg3.txt
for t in 0 3; do
CINDM_LIB_VARIANT=prof PHASE_OPTS=tune=$t timeout 300 python tools/phase_table.py cfg2 40 > $out/phase5_cfg2_tune$t.txt 2> $out/phase5_$t.err
tail -1 $out/phase5_cfg2_tune$t.txt
done


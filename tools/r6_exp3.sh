# Round 6, experiment 3 (option "tune"): 48 = exp. 2's winner (next launch's warm-up when K loop B is done, conv A tiles x and x + 8);
# +256 = every other kernel issues its warm-up late too; +64 = phase B's first weights inside pair exchange A; +128 = the warm-up
# inside pair exchange B.
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python - > $out/tune_bitwise3.txt 2>&1 <<'PY'
import torch, cindm_amd
from cindm_amd.synthetic import synthetic_init_
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.TemporalUnet1D(horizon=24, transition_dim=8, cond_dim=False, dim=64, dim_mults=(1, 2, 4, 8), attention=True), seed=0).to(dev)
d = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0, timesteps=1000, sampling_timesteps=1000).to(dev)
ref = None
for t in (0, 48, 304, 112, 176, 496):
    m.set_option("tune", t)
    x = d.sample(batch_size=256, seed=1, t_stop=960, n_composed=0, compose_n_bodies=2)
    torch.cuda.synchronize()
    if ref is None: ref = x.clone()
    print("tune", t, "bitwise equal to tune 0:", bool(torch.equal(x, ref)), "finite", bool(torch.isfinite(x).all()), flush=True)
PY
cat $out/tune_bitwise3.txt
for t in 48 256 304 112 176 240 496; do python tools/ab1d.py tune 0 $t 600 cfg2 | grep us/step; done > $out/ab_tune3.txt 2>&1
cat $out/ab_tune3.txt
for t in 0 496; do
CINDM_LIB_VARIANT=prof PHASE_SPLIT_NT=1 PHASE_OPTS=tune=$t timeout 300 python tools/phase_table.py cfg2 40 > $out/phase3_cfg2_tune$t.txt 2> $out/phase3_$t.err
tail -1 $out/phase3_cfg2_tune$t.txt
done

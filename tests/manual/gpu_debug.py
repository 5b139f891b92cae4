"""Verbose GPU bring-up check (run through gpurun): localises the first diverging block of the
HIP U-Net against the CPU oracle, then checks single steps / a chain against tests/golden."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cindm_oracle as O          # noqa: E402
import cindm_amd                   # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def build(hz, F, attention=True, seed=0):
    sd = O.synth_state_dict(O.unet1d_param_shapes(hz, F, attention=attention), seed=seed)
    m = cindm_amd.TemporalUnet1D(hz, F, False, attention=attention)
    m.load_state_dict(sd, strict=True)
    return m.to(dev), sd


def main():
    print(torch.cuda.get_device_name(0), flush=True)
    m, sd = build(24, 8)
    g = torch.Generator().manual_seed(11)
    for B, t in ((4, 500), (5, 999), (1, 0), (50, 10)):
        x = torch.randn((B, 24, 8), generator=g)
        taps = {}
        ref = O.unet1d_forward(sd, x, torch.full((B,), t, dtype=torch.long), taps=taps)
        out = m(x.to(dev), torch.full((B,), t, device=dev))
        torch.cuda.synchronize()
        print(f"B={B} t={t} eps rel err {rel(out, ref):.3e}", flush=True)
        if B == 4 or rel(out, ref) > 1e-4:
            for k, v in taps.items():
                if k in ("temb", "mid"):
                    continue
                try:
                    tv = m.tap(k, B)
                    print(f"   tap {k:14s} {tuple(tv.shape)} rel {rel(tv, v):.3e}")
                except Exception as e:
                    print(f"   tap {k}: {e}")
    for (hz, F, att) in ((24, 4, True), (24, 16, True), (24, 8, False), (44, 8, True), (8, 8, True)):
        mm, sdd = build(hz, F, att, seed=1 if F == 4 else 0)
        x = torch.randn((3, hz, F), generator=g)
        ref = O.unet1d_forward(sdd, x, torch.full((3,), 321, dtype=torch.long))
        out = mm(x.to(dev), torch.full((3,), 321, device=dev))
        print(f"hz={hz} F={F} att={att} eps rel err {rel(out, ref):.3e}", flush=True)

    # golden single steps
    st = np.load(os.path.join(ROOT, "tests", "golden", "steps_1d.npz"))
    gd = cindm_amd.GaussianDiffusion1D(m, image_size=24, conditioned_steps=0).to(dev)
    kw = dict(compose_mode="mean", n_composed=0, compose_start_step=4, single_model_step=24, compose_n_bodies=2)
    for t in (999, 500, 1, 0):
        x = torch.from_numpy(st[f"cfg2_outside_mean.t{t}.x"]).to(dev)
        nz = torch.from_numpy(st[f"cfg2_outside_mean.t{t}.noise"]).to(dev)
        out, x0 = gd.p_sample_compose_outside(x, None, t, noise=nz, **kw)
        print(f"step cfg2 t={t}: x {rel(out, torch.from_numpy(st[f'cfg2_outside_mean.t{t}.out'])):.3e} "
              f"x0 {rel(x0, torch.from_numpy(st[f'cfg2_outside_mean.t{t}.x0'])):.3e}", flush=True)
    for mode in ("mean-inside", "sum-inside"):
        kw3 = dict(compose_mode=mode, n_composed=2, compose_start_step=16, single_model_step=24, compose_n_bodies=2)
        for t in (999, 500, 1, 0):
            x = torch.from_numpy(st[f"cfg3_{mode}.t{t}.x"]).to(dev)
            nz = torch.from_numpy(st[f"cfg3_{mode}.t{t}.noise"]).to(dev)
            out, x0 = gd.p_sample_compose_inside(x, None, t, noise=nz, **kw3)
            print(f"step cfg3 {mode} t={t}: x {rel(out, torch.from_numpy(st[f'cfg3_{mode}.t{t}.out'])):.3e} "
                  f"x0 {rel(x0, torch.from_numpy(st[f'cfg3_{mode}.t{t}.x0'])):.3e}", flush=True)

    # chain cfg1: B=4, 1000 steps, explicit tape
    cpath = os.path.join(ROOT, "tests", "golden", "chains_1d.npz")
    ch = np.load(cpath) if os.path.exists(cpath) else None
    tape = O.NoiseTape.make(1234, (4, 24, 8), 1000)
    nt = cindm_amd.NoiseTape(tape.init, tape.step)
    outs = []
    for use_graph in (False, True):
        t0 = time.time()
        out = gd.sample(batch_size=4, n_composed=0, compose_n_bodies=2, noise=nt, use_graph=use_graph)
        torch.cuda.synchronize()
        outs.append(out)
        r = rel(out, torch.from_numpy(ch['cfg1.final'])) if ch is not None else float('nan')
        print(f"chain cfg1 graph={use_graph}: rel {r:.3e} ({time.time() - t0:.2f}s) finite={bool(torch.isfinite(out).all())}", flush=True)
    print("graph == eager:", bool(torch.equal(outs[0], outs[1])), flush=True)
    # throughput sniff: B=256
    x = torch.randn((256, 24, 8), device=dev)
    tt = torch.full((256,), 500, device=dev)
    for _ in range(3):
        m(x, tt)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        m(x, tt)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 20
    print(f"B=256 forward {dt * 1e3:.3f} ms  -> {256 * 160.38e6 / dt / 1e12:.2f} TFLOP/s, launches {m.launches_per_forward}")
    t0 = time.time()
    out = gd.sample(batch_size=256, n_composed=0, compose_n_bodies=2, seed=1)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"B=256 1000-step chain {dt:.3f}s -> {256 / dt:.1f} samples/s; finite={bool(torch.isfinite(out).all())} "
          f"absmax={float(out.abs().max()):.3f}")


if __name__ == "__main__":
    main()

"""Verbose GPU bring-up check of the 2-D airfoil path (run through gpurun): localises the first diverging block of
the HIP Unet against the CPU oracle, then checks one reverse step and times the forward."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cindm_oracle as O          # noqa: E402
import cindm_amd                   # noqa: E402

dev = torch.device("cuda:0")
TAPS = ["init_conv", "downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.1.1", "downs.1.2", "downs.1.3",
        "mid_block1", "mid_attn", "mid_block2", "ups.0.0", "ups.0.1", "ups.0.2", "ups.0.3", "ups.1.0", "ups.1.1",
        "ups.1.2", "ups.1.3", "final_res_block"]


def rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def main():
    print(torch.cuda.get_device_name(0), flush=True)
    for S in (64, 32):
        sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
        m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=S)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        g = torch.Generator().manual_seed(11)
        x = torch.randn((2, 21, S, S), generator=g)
        t = 500
        taps = {}
        ref = O.unet2d_forward(sd, x, torch.full((2,), t, dtype=torch.long), taps=taps)
        out = m(x.to(dev), t)
        torch.cuda.synchronize()
        print(f"S={S} launches/forward {m.launches_per_forward}  eps rel err {rel(out, ref):.3e}", flush=True)
        for n in TAPS:
            if n not in taps:
                continue
            try:
                v = m.tap(n, 2)
            except Exception as e:
                print("  tap", n, "missing:", e)
                continue
            print(f"  {n:18s} {tuple(v.shape)} rel {rel(v, taps[n]):.3e}", flush=True)
    # one reverse step
    sd = O.synth_state_dict_2d(O.unet2d_param_shapes(64, (1, 2), 21), 0)
    m = cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    d = cindm_amd.GaussianDiffusion(m, image_size=64, frames=6, timesteps=1000).to(dev)
    od = O.Diffusion2D(sd, image_size=64, frames=6)
    shape = (1, 2, 21, 64, 64)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 21, 64, 64), generator=g)
    nz = O.sample_noise_2d(torch.randn((1, 1, 18, 64, 64), generator=g), torch.randn((1, 2, 3, 64, 64), generator=g)).reshape(2, 21, 64, 64)
    ref, ref0 = O.p_sample_2d(od, shape, x.clone(), 500, nz)
    out, x0 = d.p_sample(shape, x.to(dev), 500, noise=nz.to(dev))
    print(f"step t=500: x rel {rel(out, ref):.3e}  x0 rel {rel(x0, ref0):.3e}", flush=True)
    # timing: forward over a batch of images, and short chains
    for NI in (2, 16, 64):
        xx = torch.randn((NI, 64 * 64, m.padded_channels), device=dev)
        for _ in range(3):
            m.forward_device_layout(xx, 500)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            m.forward_device_layout(xx, 500)
        torch.cuda.synchronize()
        print(f"forward {NI} images: {(time.time() - t0) / 10 * 1e3:.3f} ms", flush=True)
    for B, nb in ((1, 2), (16, 2)):
        d.sample(batch_size=B, num_boundaries=nb, seed=1, t_stop=990)
        torch.cuda.synchronize()
        t0 = time.time()
        d.sample(batch_size=B, num_boundaries=nb, seed=1, t_stop=900)
        torch.cuda.synchronize()
        print(f"chain B={B} nb={nb}: {(time.time() - t0) / 100 * 1e3:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()

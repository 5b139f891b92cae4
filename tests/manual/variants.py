"""debug: run one U-Net variant against the oracle (python tests/manual/variants.py hz F att B)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cindm_amd, cindm_oracle as O
hz, F, att, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
sd = O.synth_state_dict(O.unet1d_param_shapes(hz, F, attention=bool(att)), seed=0)
m = cindm_amd.TemporalUnet1D(hz, F, False, attention=bool(att))
m.load_state_dict(sd, strict=True); m = m.to(dev)
for kv in sys.argv[5:]:
    k, v = kv.split("="); m.set_option(k, int(v))
x = torch.randn((B, hz, F), generator=torch.Generator().manual_seed(3))
ref = O.unet1d_forward(sd, x, torch.full((B,), 321, dtype=torch.long))
out = m(x.to(dev), torch.full((B,), 321, device=dev))
torch.cuda.synchronize()
print(sys.argv[1:], "launches", m.launches_per_forward, "rel err", float((out.cpu() - ref).abs().max() / ref.abs().max()), flush=True)

"""2-D workspace size with and without aliasing of block-internal temporaries: python tools/wsz2d.py"""
import sys, torch
sys.path.insert(0, "/root/repo"); import cindm_amd
from cindm_amd.synthetic import synthetic_init_
from cindm_amd import _ffi
dev = torch.device("cuda:0")
m = synthetic_init_(cindm_amd.Unet(dim=64, dim_mults=(1, 2), channels=21, image_size=64), 0).to(dev)
m.sync_weights()
for v in (0, 1):
    m.set_option("ws_alias", v); m.sync_weights()
    print("ws_alias", v, "workspace for 128 images: %.1f MB" % (_ffi.lib().cindm_unet2d_workspace_bytes(m._h, 128) / 1e6))

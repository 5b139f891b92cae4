"""ctypes binding of libcindm_hip.so (C ABI: include/cindm_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing the
product classes raises.  PyTorch is used only for device memory (raw ``data_ptr()``), the
current HIP stream and ``torch.distributed``.
"""
import ctypes as C
import os

from . import build as _build

_lib = None
ABI_VERSION = 4          # == CINDM_ABI_VERSION of include/cindm_hip.h (tests/test_host_logic.py keeps the two in step)


class CindmError(RuntimeError):
    pass


class UnetDesc(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("transition_dim", C.c_int32), ("dim", C.c_int32),
                ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 8), ("attention", C.c_int32),
                ("timesteps", C.c_int32)]


class Unet2dDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 4), ("channels", C.c_int32),
                ("image_size", C.c_int32), ("timesteps", C.c_int32)]


SCHED_NAMES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
               "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
               "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
               "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
               "loss_weight")


class ForceUnetDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 4), ("channels", C.c_int32),
                ("image_size", C.c_int32)]


class SchedDesc(C.Structure):
    _fields_ = [("timesteps", C.c_int32)] + [(n, C.c_void_p) for n in SCHED_NAMES]


class ComposeDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_windows", C.c_int32), ("compose_start_step", C.c_int32),
                ("window", C.c_int32), ("n_bodies", C.c_int32), ("cond_steps", C.c_int32),
                ("objective", C.c_int32), ("clip_denoised", C.c_int32), ("uncond_coef", C.c_float)]


class DesignDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("alpha", C.c_int32), ("recurrence", C.c_int32), ("last_n_step", C.c_int32),
                ("coef", C.c_float), ("time_consistency_coef", C.c_float), ("pos_target", C.c_float * 2)]


COMPOSE_PLAIN, COMPOSE_MEAN_INSIDE, COMPOSE_SUM_INSIDE, COMPOSE_MEAN_OUTSIDE, COMPOSE_NOISESUM_OUTSIDE, \
    COMPOSE_MULTIBODY = range(6)
OBJECTIVES = {"pred_noise": 0, "pred_x0": 1, "pred_v": 2}

_vp, _i32, _i64, _u64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); every symbol include/cindm_hip.h declares
SIGNATURES = {
    "cindm_abi_version": (C.c_int, []),
    "cindm_last_error": (C.c_char_p, []),
    "cindm_source_hash": (C.c_char_p, []),
    "cindm_ws_prof_read": (C.c_int, [_vp]),
    "cindm_ddpm1d_last_chain_info": (C.c_int, [_vp, C.POINTER(_i32)]),
    "cindm_forceunet_status": (C.c_int, [_vp, _vp]),
    "cindm_forceunet_recovered": (C.c_int, [_vp]),
    "cindm_comm_unique_id": (C.c_int, [_vp]),
    "cindm_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "cindm_comm_world": (C.c_int, [_vp]),
    "cindm_all_gather_designs": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "cindm_comm_destroy": (None, [_vp]),
    "cindm_unet1d_status": (C.c_int, [_vp, _vp]),
    "cindm_unet1d_poll": (C.c_int, [_vp, _vp]),
    "cindm_unet1d_recovered": (C.c_int, [_vp]),
    "cindm_unet1d_range_escalate": (C.c_int, [_vp, _i32, _vp]),
    "cindm_unet1d_phase_prof_enable": (C.c_int, [_vp, _i32]),
    "cindm_unet1d_phase_prof_read": (C.c_int, [_vp, _vp, _i64, _vp]),
    "cindm_unet1d_phase_prof_name": (C.c_char_p, [_vp, _i32]),
    "cindm_unet1d_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "cindm_unet1d_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "cindm_unet2d_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "cindm_unet2d_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "cindm_unet1d_create": (C.c_int, [C.POINTER(UnetDesc), C.POINTER(_vp)]),
    "cindm_unet1d_destroy": (None, [_vp]),
    "cindm_unet1d_num_params": (C.c_int, [_vp]),
    "cindm_unet1d_param_info": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(_i64 * 4), C.POINTER(C.c_int)]),
    "cindm_unet1d_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64, C.c_int]),
    "cindm_unet1d_set_sinusoid_table": (C.c_int, [_vp, _vp, _i64]),
    "cindm_unet1d_finalize": (C.c_int, [_vp, _vp]),
    "cindm_unet1d_workspace_bytes": (_sz, [_vp, _i64]),
    "cindm_unet1d_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "cindm_unet1d_tap": (C.c_int, [_vp, C.c_char_p, _i64, _vp, _vp, _i64, C.POINTER(_i64 * 3), _vp]),
    "cindm_unet1d_profile": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _sz, _vp, C.POINTER(_i32 * 6),
                                       C.POINTER(C.c_float * 6), C.POINTER(C.c_double * 6)]),
    "cindm_unet1d_profile_detail": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _sz, _vp, _i32, C.POINTER(_i32), _vp, _vp, _vp, _vp]),
    "cindm_unet1d_launches_per_forward": (C.c_int, [_vp]),
    "cindm_ddpm1d_create": (C.c_int, [C.POINTER(SchedDesc), C.POINTER(_vp)]),
    "cindm_ddpm1d_destroy": (None, [_vp]),
    "cindm_ddpm1d_workspace_bytes": (_sz, [_vp, _vp, _vp, C.POINTER(ComposeDesc), _i64]),
    "cindm_ddpm1d_predict": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc), _vp, _vp, _i32, _vp, _i64,
                                       _vp, _vp, _vp, _vp, _sz, _vp]),
    "cindm_ddpm1d_step": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc), _vp, _vp, _vp, _u64, _i64,
                                    _vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "cindm_ddpm1d_sample": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc), _vp, _vp, _vp, _u64, _i64,
                                      _vp, _i32, _vp, _i32, _i32, _i64, _vp, _sz, _vp, _i32]),
    "cindm_ddpm1d_sample_ddim": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc), _vp, _vp, _i32, _vp, _vp, _vp, _u64, _i64,
                                           _vp, _i32, _vp, _i64, _vp, _sz, _vp, _i32]),
    "cindm_ddpm1d_sample_guided": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc), C.POINTER(DesignDesc), _vp, _vp, _vp, _vp,
                                             _u64, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _sz, _vp, _i32]),
    "cindm_fill_normal": (C.c_int, [_vp, _i64, _i64, _u64, _i64, _i32, _vp]),
    "cindm_ddpm1d_launches_per_step": (C.c_int, [_vp, _vp, _vp, C.POINTER(ComposeDesc)]),
    "cindm_ddpm1d_last_step_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "cindm_unet2d_create": (C.c_int, [C.POINTER(Unet2dDesc), C.POINTER(_vp)]),
    "cindm_unet2d_destroy": (None, [_vp]),
    "cindm_unet2d_num_params": (C.c_int, [_vp]),
    "cindm_unet2d_param_info": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(_i64 * 4), C.POINTER(C.c_int)]),
    "cindm_unet2d_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64, C.c_int]),
    "cindm_unet2d_set_sinusoid_table": (C.c_int, [_vp, _vp, _i64]),
    "cindm_unet2d_finalize": (C.c_int, [_vp, _vp]),
    "cindm_unet2d_padded_channels": (C.c_int, [_vp]),
    "cindm_unet2d_workspace_bytes": (_sz, [_vp, _i64]),
    "cindm_unet2d_launches_per_forward": (C.c_int, [_vp]),
    "cindm_unet2d_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "cindm_unet2d_profile": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _sz, _vp, C.POINTER(_i32 * 7),
                                       C.POINTER(C.c_float * 7), C.POINTER(C.c_double * 7)]),
    "cindm_unet2d_tap": (C.c_int, [_vp, C.c_char_p, _i64, _vp, _vp, _i64, C.POINTER(_i64 * 3), _vp]),
    "cindm_ddpm2d_workspace_bytes": (_sz, [_vp, _i64]),
    "cindm_ddpm2d_step": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _u64, _i64, _i32, _vp, _vp, _vp,
                                    _vp, _sz, _vp]),
    "cindm_ddpm2d_sample": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _u64, _i64, _i32, _i32, _vp, _sz, _vp,
                                      _i32]),
    "cindm_ddpm2d_predict": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cindm_fill_noise2d": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _u64, _i64, _i32, _vp]),
    "cindm_forceunet_create": (C.c_int, [C.POINTER(ForceUnetDesc), C.POINTER(_vp)]),
    "cindm_forceunet_destroy": (None, [_vp]),
    "cindm_forceunet_num_params": (C.c_int, [_vp]),
    "cindm_forceunet_param_info": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(_i64 * 4), C.POINTER(C.c_int)]),
    "cindm_forceunet_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64, C.c_int]),
    "cindm_forceunet_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "cindm_forceunet_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "cindm_forceunet_finalize": (C.c_int, [_vp, _vp]),
    "cindm_forceunet_workspace_bytes": (_sz, [_vp, _i64, _i32]),
    "cindm_forceunet_forward": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "cindm_forceunet_grad": (C.c_int, [_vp, _vp, C.c_float, _vp, _vp, _i64, _vp, _sz, _vp]),
    "cindm_forceunet_vjp": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "cindm_airfoil_design_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32]),
    "cindm_airfoil_design_grad": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float, _i32,
                                            _i32, _vp, _vp, _sz, _vp]),
    "cindm_ddpm2d_sample_force": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _u64, _i64, _i32, _i32, _i32,
                                            C.c_float, C.c_float, C.c_float, C.c_float, _i32, _i32, _vp, _vp, _vp, _sz, _vp, _sz,
                                            _vp, _i32]),
}


def lib():
    """Loads libcindm_hip.so, building it first when it is missing or was compiled from other sources than the ones
    next to it (the library embeds the sha256 of its sources).  A stale library is never used silently."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()
    if _build.needs_build():
        if os.environ.get("LOCAL_RANK", "0") not in ("0", ""):
            # multi-process launch (torchrun): local rank 0 builds, the others wait for its library
            import time
            for _ in range(1200):
                if not _build.needs_build():
                    break
                time.sleep(0.5)
        else:
            try:
                _build.build()
            except Exception as e:
                raise CindmError(f"libcindm_hip.so is missing or stale and could not be built: {e}") from e
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise CindmError(f"cannot load {path}: {e} -- the HIP extension is required, there is no CPU fallback") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.cindm_abi_version() != ABI_VERSION:
        raise CindmError(f"libcindm_hip.so ABI version {L.cindm_abi_version()}, this binding is written for {ABI_VERSION}")
    have, want = L.cindm_source_hash().decode(), _build.source_hash()
    if have != want:
        raise CindmError(f"libcindm_hip.so was built from other sources (library {have[:12]}, tree {want[:12]}); "
                         "run `python -m cindm_amd.build --force`")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise CindmError(lib().cindm_last_error().decode())


def ptr(t):
    """Raw device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

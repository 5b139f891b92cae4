"""In-replay phase clocks of the 1-D reverse step (profiling build of the library: CINDM_LIB_VARIANT=prof).

    CINDM_LIB_VARIANT=prof python tools/phase_table.py [cfg2|cfg3|cfg4] [steps] > profiles/r04_phase_table_<wl>.txt

Runs `steps` replayed reverse steps of the bench workload with the phase clocks armed (csrc/kernels.h PhaseBuf: every wave of
every workgroup stamps the 100 MHz constant clock at its phase boundaries and writes the stamps at the end of the kernel), reads
the records of the LAST step and prints, per launch of that step:
  * span     first workgroup's entry -> last workgroup's last stamp (what a kernel trace calls the kernel's duration, minus the
             dispatch ramp and the end-of-kernel drain)
  * gap      previous launch's last stamp -> this launch's first entry (drain + dependent dispatch)
  * skew     first -> last workgroup entry
  * per phase: median over all (workgroup, wave) of the time between consecutive stamps, and the same for the workgroup that
             finished LAST (the critical one).  The medians of a launch add up to its median workgroup lifetime.
The clock ticks every 10 ns: single phases below ~0.1 us are resolution-limited, sums are not.
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("CINDM_LIB_VARIANT") in ("prof", "abl1", "abl2", "abl3", "kprof"), "run with CINDM_LIB_VARIANT=prof (the production library has no phase clocks)"
import bench                                 # noqa: E402
from cindm_amd import _ffi                   # noqa: E402

MAXWG, MAXWAVE, NST = 1024, 8, 16
TICK_US = 0.01

PHASES = {
    "dconv2": ["entry", "prologue loads issued", "K loop A (tile wait + MFMA)", "cross-wave reduce A", "GN stats A", "pair exchange A",
               "Mish + planes + sc1 stores issued", "store drain + flag", "1x1 residual reduce", "poll producers' flags", "K loop B (y0 fetch + MFMA)",
               "cross-wave reduce B", "GN stats B", "pair exchange B", "Mish + residual + stores issued"],
    "dresample": ["entry", "loads issued", "K loop", "cross-wave reduce", "planes + stores issued"],
    "attn1d_head": ["entry", "weight loads issued", "LayerNorm -> planes", "q|k|v K share (MFMA)", "cross-wave sum", "core",
                    "tile published", "four heads gathered (spin)", "att planes in LDS", "out projection + stores issued"],
    "attn1d_site": ["entry", "rows + weights requested, LayerNorm -> planes", "q|k|v + core + att planes", "out projection + stores issued"],
    "level0_down": ["entry", "stage x + zero fill", "b0.conv0 + residual_conv", "b0.conv1", "b1.conv0", "b1.conv1 -> h2", "LayerNorm",
                    "q|k|v + core", "out projection", "downsample + stores"],
    "level1_down": ["entry", "zero fill", "stage x", "b0.conv0 + residual_conv", "b0.conv1", "b1.conv0", "b1.conv1 -> h2", "LayerNorm",
                    "q|k|v + core (H free)", "att planes", "out projection + skip store", "downsample + stores"],
    "ups_last": ["entry"] + [f"after barrier {k}" for k in range(1, 12)] + ["final stores (+ fused DDPM update)"],
    "ups_tail128": ["entry"] + [f"after barrier {k}" for k in range(1, 9)] + ["final stores"],
}


if os.environ.get("CINDM_LIB_VARIANT", "").startswith("kprof"):      # the experiment build with the clocks inside phase A's K loop (kernels_dconv.h, CINDM_KPROF)
    PHASES["dconv2"] = ["entry", "prologue loads issued"] + [x for j in range(4) for x in (f"k-step {j} staged (tile wait, LDS write)", f"k-step {j} multiplied (5 taps)")]


def kind_of(name):
    return name.split("<")[0].split(" ")[0]


def read(model, dev):
    L = _ffi.lib()
    cap = 32 * MAXWG * MAXWAVE * NST
    buf = np.zeros(cap, dtype=np.uint64)
    with torch.cuda.device(dev):
        n = L.cindm_unet1d_phase_prof_read(model._h, buf.ctypes.data_as(C.c_void_p), cap, _ffi.current_stream(dev))
    if n < 0:
        raise RuntimeError(L.cindm_last_error().decode())
    names = [L.cindm_unet1d_phase_prof_name(model._h, i).decode() for i in range(n)]
    return names, buf[:n * MAXWG * MAXWAVE * NST].reshape(n, MAXWG, MAXWAVE, NST).astype(np.int64)


def table(names, rec, out):
    prev_end = None
    tot_span = tot_gap = 0.0
    for s, name in enumerate(names):
        r = rec[s]
        used = r[:, :, 0] != 0                      # (wg, wave) pairs that ran
        if not used.any():
            continue
        wg_used = used.any(axis=1)
        r = r[wg_used]
        u = used[wg_used]
        start = np.where(u, r[:, :, 0], np.iinfo(np.int64).max).min()
        last = r.max(axis=2)                        # per (wg, wave): the latest stamp
        end = last[u].max()
        entry = np.where(u, r[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
        span, skew = (end - start) * TICK_US, (entry.max() - start) * TICK_US
        gap = (start - prev_end) * TICK_US if prev_end is not None else float("nan")
        prev_end = end
        tot_span += span
        if gap == gap:
            tot_gap += gap
        if os.environ.get("PHASE_RAW") == kind_of(name):      # every stamp as its median distance from the entry stamp (marks that are not in program order)
            out.write(f"\n#{s:2d} {name}: stamps relative to entry (us, median): " + " ".join(
                f"[{i}] {np.median((r[:, :, i] - r[:, :, 0])[u & (r[:, :, i] != 0)]) * TICK_US:.2f}" for i in range(1, NST) if (u & (r[:, :, i] != 0)).any()) + "\n")
        labels = PHASES.get(kind_of(name), [])
        crit_wg = int(np.argmax(np.where(u, last, 0).max(axis=1)))
        out.write(f"\n#{s:2d} {name}: {int(wg_used.sum())} workgroups x {int(u[0].sum())} waves | span {span:6.2f} us | gap before {gap:5.2f} us | entry skew {skew:5.2f} us\n")
        out.write(f"    {'phase':46s} {'median':>8s} {'p90':>8s} {'last WG':>8s}   (us)\n")
        msum = csum = 0.0
        prev_i = 0
        for i in range(1, NST):
            cur, prv = r[:, :, i], r[:, :, prev_i]
            ok = u & (cur != 0) & (prv != 0)
            if not ok.any():
                continue
            d = (cur - prv)[ok] * TICK_US
            c = (r[crit_wg, :, i] - r[crit_wg, :, prev_i])[ok[crit_wg]] * TICK_US
            lab = labels[i] if i < len(labels) else f"stamp {i}"
            cm = float(np.median(c)) if c.size else float("nan")
            out.write(f"    {lab:46s} {np.median(d):8.2f} {np.percentile(d, 90):8.2f} {cm:8.2f}\n")
            msum += float(np.median(d)); csum += cm if cm == cm else 0.0
            prev_i = i
        out.write(f"    {'sum of phases (workgroup lifetime)':46s} {msum:8.2f} {'':8s} {csum:8.2f}\n")
        if os.environ.get("PHASE_SPLIT_NT") and kind_of(name) == "dconv2":
            # n-tiles 0-7 have their fragments warmed by the predecessor launch (kernels.h Pf: two regions), 8-15 do not
            NT = 4 * int(name.split("<")[1].split(">")[0].split(",")[-1])
            wg_ids = np.nonzero(wg_used)[0]
            for lab, sel in (("n-tiles 0-7", (wg_ids % NT) < 8), ("n-tiles 8-15", (wg_ids % NT) >= 8)):
                if not sel.any():
                    continue
                parts = []
                prev_i = 0
                for i in range(1, NST):
                    cur, prv = r[sel][:, :, i], r[sel][:, :, prev_i]
                    ok = u[sel] & (cur != 0) & (prv != 0)
                    if not ok.any():
                        continue
                    parts.append(f"{np.median((cur - prv)[ok]) * TICK_US:.2f}")
                    prev_i = i
                out.write(f"    medians, {lab:13s}: " + " ".join(parts) + "\n")
    out.write(f"\nsum over the step's instrumented launches: spans {tot_span:.1f} us + gaps {tot_gap:.1f} us = {tot_span + tot_gap:.1f} us\n")


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = bench.default_batch(wl)
    w = bench.build_1d(wl, B, dev)
    d, model = w["diffusion"], w["pair"]
    for kv in filter(None, os.environ.get("PHASE_OPTS", "").split(",")):      # e.g. PHASE_OPTS=tune=4,l2_prefetch=0
        k, v = kv.split("=")
        model.set_option(k, int(v))
    if wl == "cfg4":
        run = lambda n: d.sample_compose_multibodies(w["cond"], n, 0, 4, seed=1)
    else:
        kw = dict(n_composed=0, compose_n_bodies=2)
        kw.update(w.get("compose_kw", {}))
        run = lambda n: d.sample(batch_size=B, cond=None, seed=1, t_stop=1000 - n, **kw)
    run(4)
    torch.cuda.synchronize()
    t0 = time.time(); run(steps); torch.cuda.synchronize()
    base = (time.time() - t0) / steps * 1e6
    model.sync_weights()
    with torch.cuda.device(dev):
        _ffi.check(_ffi.lib().cindm_unet1d_phase_prof_enable(model._h, 1))
    run(4)
    torch.cuda.synchronize()
    t0 = time.time(); run(steps); torch.cuda.synchronize()
    armed = (time.time() - t0) / steps * 1e6
    names, rec = read(model, dev)
    out = sys.stdout
    out.write(f"# tools/phase_table.py {wl} {steps}: library {_ffi.lib().cindm_source_hash().decode()[:12]} (profiling build, -DCINDM_PHASE_PROF)\n")
    out.write(f"# reverse step, host-timed over {steps} replayed steps: {base:.1f} us with the clocks compiled in but not armed, {armed:.1f} us armed\n")
    out.write("# (the production build of the same sources is what bench.py times; the difference is the instrumentation's cost)\n")
    table(names, rec, out)


if __name__ == "__main__":
    main()

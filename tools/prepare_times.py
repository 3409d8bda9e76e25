#!/usr/bin/env python3
"""The stack's preparation launch (txe_gat_layers_prepare) stand-alone on the BASELINE batch shape: both layers, each alone, and with
the feature dropout off -- which of its jobs is the long pole:  gpurun -- python tools/prepare_times.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib  # noqa: E402
from taxoexpan_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
N = 17877
h = torch.randn(N, 250, device=dev)
pos = torch.randint(0, 3, (N,), dtype=torch.int32, device=dev)


def layer(Kh, Pd, H, D, has_h, p):
    Kp, Fp = call("txe_gat_padded_k", Kh, Pd), call("txe_gat_padded_f", H, D)
    d = dict(X=torch.empty(N, Kp, device=dev), W=torch.randn(H * D, Kh + Pd, device=dev), al=torch.randn(H * D, device=dev), ar=torch.randn(H * D, device=dev),
             P=torch.randn(3, Pd, device=dev), Wp=torch.empty(Fp, Kp, device=dev), mask=torch.empty(N, (Kh + Pd + 31) // 32, dtype=torch.int32, device=dev),
             Kh=Kh, Pd=Pd, H=H, D=D, has_h=has_h, p=p)
    return d


def run(layers):
    descs = (_lib.GatPrepareDesc * len(layers))()
    for d, L in zip(descs, layers):
        d.h, d.ld_h, d.n_nodes, d.Kh, d.pos, d.P, d.Pd, d.X = (ptr(h) if L["has_h"] else None), (250 if L["has_h"] else 0), N, L["Kh"], ptr(pos), ptr(L["P"]), L["Pd"], ptr(L["X"])
        d.W, d.attn_l, d.attn_r, d.H, d.D, d.Wp = ptr(L["W"]), ptr(L["al"]), ptr(L["ar"]), L["H"], L["D"], ptr(L["Wp"])
        d.feat_drop_p, d.seed, d.mask = L["p"], 1234, ptr(L["mask"]) if L["p"] > 0 else None
        d.x_dropped = int(L["has_h"] and L["p"] > 0 and not L.get("no_xd"))
    call("txe_gat_layers_prepare", ctypes.cast(descs, ctypes.c_void_p), len(layers), _lib.stream_ptr())


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for p in (0.1, 0.25, 0.5, 0.0):
    L0, L1 = layer(250, 50, 4, 500, True, p), layer(2000, 50, 1, 500, False, p)
    L0n = dict(L0, no_xd=True)
    print(f"p={p}: layer 0 alone, mask job + plain build_x {t(lambda: run([L0n])):.1f} us")
    print(f"p={p}: both {t(lambda: run([L0, L1])):.1f} us   layer 0 alone {t(lambda: run([L0])):.1f} us   layer 1 alone {t(lambda: run([L1])):.1f} us")

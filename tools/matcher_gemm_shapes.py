#!/usr/bin/env python3
"""The matcher's two 1-GFLOP products on the BASELINE batch (G = 4,096 pairs, l = 500, r = 250) with the operands as they come (rows of
250 floats: 8-byte vector loads at best) against rows padded to 256 floats (16-byte loads), stand-alone.
    python tools/matcher_gemm_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib  # noqa: E402
from taxoexpan_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
G, l, r, rp = 4096, 500, 250, 256
torch.manual_seed(0)
e2 = torch.randn(G, r, device=dev)
W = torch.randn(l, r, device=dev)
R = torch.randn(G, l, device=dev)
e2p = torch.zeros(G, rp, device=dev); e2p[:, :r] = e2
Wp = torch.zeros(l, rp, device=dev); Wp[:, :r] = W
tail = torch.empty(call("txe_gemm_tail_ws_bytes"), dtype=torch.uint8, device=dev)


def t(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


V = torch.empty(G, l, device=dev)
st = _lib.stream_ptr()
print("V = E2 W^T   rows of 250: %.1f us" % t(lambda: call("txe_gemm_plain", 0, ptr(e2), r, ptr(W), r, ptr(V), l, G, l, r, 1, ptr(tail), tail.numel(), st)))
V2 = torch.empty(G, l, device=dev)
print("V = E2 W^T   rows of 256: %.1f us" % t(lambda: call("txe_gemm_plain", 0, ptr(e2p), rp, ptr(Wp), rp, ptr(V2), l, G, l, rp, 1, ptr(tail), tail.numel(), st)))
print("   equal bits:", torch.equal(V, V2))
for S in (4, 8, 16):
    part = torch.empty(S, l, r, device=dev)
    a = t(lambda: call("txe_gemm_plain", 2, ptr(R), l, ptr(e2), r, ptr(part), r, l, r, G, S, None, 0, st))
    part2 = torch.empty(S, l, r, device=dev)
    b = t(lambda: call("txe_gemm_plain", 2, ptr(R), l, ptr(e2p), rp, ptr(part2), r, l, r, G, S, None, 0, st))
    print(f"dW = R^T E2, {S:2d} slices   rows of 250: {a:.1f} us   rows of 256: {b:.1f} us   equal bits: {torch.equal(part, part2)}")

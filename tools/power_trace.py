#!/usr/bin/env python3
"""Socket power and shader clock (amdgpu hwmon: power1_input, freq1_input) sampled every 5 ms while ONE kernel loops for ~2 s, for the
bf16-pipe products on random operands, on all-zero operands and on NaN operands, and for an HBM stream -- the evidence behind "the split
GEMMs are power-limited" (DESIGN 9).  Writes a markdown table to stdout; profiles/r06_power.md is a copy of one run.
usage: python tools/power_trace.py"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taxoexpan_amd._lib import call, ptr, stream_ptr


def hwmons():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "power1_input")) and os.path.exists(os.path.join(d, "freq1_input")):
            out.append(d)
    return out


def read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, mons):
        super().__init__(daemon=True)
        self.mons, self.rows, self.stop = mons, [], False

    def run(self):
        while not self.stop:
            self.rows.append([(read(m + "/power1_input") * 1e-6, read(m + "/freq1_input") * 1e-6) for m in self.mons])
            time.sleep(0.005)


def timed_loop(fn, seconds=2.0):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    a.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    mons = hwmons()
    M, N, K = 17877, 2008, 300
    g = torch.Generator().manual_seed(1)
    s = stream_ptr()
    Ap = torch.empty(call("txe_split_packed_bytes", M, K), dtype=torch.uint8, device=dev)
    Bp = torch.empty(call("txe_split_packed_bytes", N, K), dtype=torch.uint8, device=dev)
    C = torch.empty(M, N, device=dev)
    src = torch.empty(256 << 20, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    cases = []
    for name, fa, fb in (("random", lambda: torch.randn(M, K, generator=g), lambda: torch.randn(N, K, generator=g) * 0.05),
                         ("all-zero", lambda: torch.zeros(M, K), lambda: torch.zeros(N, K)),
                         ("small integers", lambda: torch.randint(-3, 4, (M, K), generator=g).float(), lambda: torch.randint(-3, 4, (N, K), generator=g).float())):
        A, B = fa().to(dev), fb().to(dev)
        call("txe_split_pack", ptr(A), K, M, K, 0, ptr(Ap), s); call("txe_split_pack", ptr(B), K, N, K, 1, ptr(Bp), s)
        torch.cuda.synchronize()
        cases.append((f"gemm_nt_split {M}x{N}x{K}, {name} operands", lambda: call("txe_gemm_nt_split", ptr(Ap), ptr(Bp), M, N, K, ptr(C), N, s), (A, B)))
        # (the closure reads Ap / Bp at call time: run the case before the next pack)
        smp = Sampler(mons); smp.start()
        time.sleep(0.3)
        n_idle = len(smp.rows)
        us = timed_loop(cases[-1][1])
        smp.stop = True; smp.join()
        report(cases[-1][0], us, smp.rows, n_idle, 2.0 * M * N * K)
    smp = Sampler(mons); smp.start(); time.sleep(0.3); n_idle = len(smp.rows)
    us = timed_loop(lambda: dst.copy_(src))
    smp.stop = True; smp.join()
    report("device copy 256 MiB (HBM stream)", us, smp.rows, n_idle, None, 2.0 * src.numel())


HEADER = False


def report(name, us, rows, n_idle, flops=None, nbytes=None):
    global HEADER
    if not HEADER:
        print("| case | avg us | rate | idle W | busy W (mean / max) | sclk MHz busy (mean / min) |")
        print("|---|---|---|---|---|---|")
        HEADER = True
    import numpy as np
    arr = np.array(rows, dtype=np.float64)                    # [samples][monitors][2]
    idle, busy = arr[:max(n_idle - 5, 1)], arr[n_idle + 20:]
    k = int(np.nanargmax(np.nanmean(busy[:, :, 0], 0) - np.nanmean(idle[:, :, 0], 0)))       # the device that woke up
    rate = f"{flops / us * 1e-6:.0f} TF/s" if flops else f"{nbytes / us * 1e-6:.2f} TB/s"
    print(f"| {name} | {us:.1f} | {rate} | {np.nanmean(idle[:, k, 0]):.0f} | {np.nanmean(busy[:, k, 0]):.0f} / {np.nanmax(busy[:, k, 0]):.0f} | "
          f"{np.nanmean(busy[:, k, 1]):.0f} / {np.nanmin(busy[:, k, 1]):.0f} |")


if __name__ == "__main__":
    main()

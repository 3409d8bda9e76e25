"""CPU restatement of the arithmetic behind csrc/txe_gemm_split.*  -- no GPU: an fp32 number is the EXACT sum of three bf16
numbers, a product of two bf16 numbers is exact in fp32, the three plane products the kernels drop are at most 2^-23 |a b| and 2^-27 |a b| in
the root mean square (an fp32 multiply's own rounding: at most 2^-24, 2^-25 rms), and a dot
product summed from the six kept ones in fp32 is as close to the float64 value as an fp32 dot product; the slot permutations of the
packed operands are bijections."""
import numpy as np


def bf16_rne(x):
    """float32 -> float32 holding the nearest bf16 (ties to even): what v_cvt_pk_bf16_f32 does for finite inputs"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x1 = bf16_rne(x)
    r1 = (x - x1).astype(np.float32)
    x2 = bf16_rne(r1)
    r2 = (r1 - x2).astype(np.float32)
    x3 = bf16_rne(r2)
    return x1, x2, x3


def _samples(n, seed):
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal(n) * np.exp(rs.uniform(-20, 20, n))).astype(np.float32)
    x[:8] = [0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, np.float32(np.pi), 2.0 ** 100, -2.0 ** -100]
    return x


def test_three_bf16_numbers_carry_an_fp32_number_exactly():
    x = _samples(200000, 1)
    x1, x2, x3 = split3(x)
    for p in (x1, x2, x3):
        assert not (p.view(np.uint32) & 0xFFFF).any()                       # bf16 values
    np.testing.assert_array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))
    # the residuals the device code forms in fp32 are exact
    np.testing.assert_array_equal((x - x1).astype(np.float64), x.astype(np.float64) - x1.astype(np.float64))
    nz = x != 0
    assert (np.abs(x2[nz].astype(np.float64)) <= 2.0 ** -8 * np.abs(x[nz].astype(np.float64))).all()
    assert (np.abs(x3[nz].astype(np.float64)) <= 2.0 ** -16 * np.abs(x[nz].astype(np.float64))).all()


def test_plane_products_are_exact_in_fp32_and_the_dropped_ones_are_a_fraction_of_one_rounding():
    a, b = _samples(100000, 2)[8:], _samples(100000, 3)[8:]
    a *= np.float32(1e-6)                                                    # (keep every product inside fp32's range)
    pa, pb = split3(a), split3(b)
    for i in range(3):
        for j in range(3):
            p32 = (pa[i] * pb[j]).astype(np.float32)
            np.testing.assert_array_equal(p32.astype(np.float64), pa[i].astype(np.float64) * pb[j].astype(np.float64))
    exact = a.astype(np.float64) * b.astype(np.float64)
    kept = sum(pa[i].astype(np.float64) * pb[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)))
    rel = np.abs(exact - kept) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23                                           # worst case: |a2| <= 2^-8 |a|, |b3| <= 2^-16 |b|, twice
    r32 = np.abs(exact - (a * b).astype(np.float32).astype(np.float64)) / np.abs(exact)     # what ONE fp32 multiply loses
    assert np.sqrt((rel ** 2).mean()) <= 0.3 * np.sqrt((r32 ** 2).mean())    # rms 2^-27.4 against 2^-25.2


def test_dot_products_from_six_plane_products_match_fp32_accuracy():
    rs = np.random.RandomState(4)
    K, n = 304, 400
    A = rs.standard_normal((n, K)).astype(np.float32)
    B = (rs.standard_normal((n, K)) * 0.05).astype(np.float32)
    pa, pb = split3(A), split3(B)
    acc = np.zeros(n, dtype=np.float32)
    for k0 in range(0, K, 16):                                               # the kernel's order: per k-tile, small products first
        for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
            blk = (pa[i][:, k0:k0 + 16].astype(np.float64) * pb[j][:, k0:k0 + 16].astype(np.float64)).sum(1)   # a 16-deep MFMA block
            acc = (acc.astype(np.float64) + blk).astype(np.float32)          # one fp32 accumulator rounding per instruction
    f32 = np.zeros(n, dtype=np.float32)
    for k in range(K):                                                       # an fp32 fma chain (the fp32 MFMA's arithmetic)
        f32 = (f32.astype(np.float64) + A[:, k].astype(np.float64) * B[:, k].astype(np.float64)).astype(np.float32)
    ref = (A.astype(np.float64) * B.astype(np.float64)).sum(1)
    scale = (np.abs(A).astype(np.float64) * np.abs(B).astype(np.float64)).sum(1)
    e_split, e_f32 = (np.abs(acc - ref) / scale).max(), (np.abs(f32 - ref) / scale).max()
    assert e_split <= e_f32 * 1.5 and e_split < 2e-7, (e_split, e_f32)


def test_slot_permutations_are_bijections():
    # csrc/txe_gemm_split.h split_slot_row, side 1: fragment rb, slot s -> column of a 128-column tile
    cols = sorted(128 * (rb >> 2) + 64 * ((rb & 3) >> 1) + 2 * s + (rb & 1) for rb in range(8) for s in range(32))
    assert cols == list(range(256))
    # the contraction-major operand's 160-column tile: blocks j < 4: 4 s + j, block 4: 128 + s
    cols = sorted([4 * s + j for j in range(4) for s in range(32)] + [128 + s for s in range(32)])
    assert cols == list(range(160))
    # the TN product's A side: slot s of block fb -> column 64 (fb >> 1) + 2 s + (fb & 1) of the 128-column tile
    cols = sorted(64 * (fb >> 1) + 2 * s + (fb & 1) for fb in range(4) for s in range(32))
    assert cols == list(range(128))

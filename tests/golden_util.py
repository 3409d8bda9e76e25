"""Helpers shared by the golden / parity tests (test infrastructure)."""
import os

import numpy as np
import torch

import golden_cases as gc
import txe_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    spec = gc.CASES[name]
    z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    shapes, x, q = gc.make_inputs(spec)
    params = gc.make_params(spec)
    if spec["full"]:  # full cases also carry inputs/params: check the seed regeneration is bit-stable
        assert np.array_equal(z["x"], x) and np.array_equal(z["q"], q)
        for k, v in params.items():
            assert np.array_equal(z["param:" + k], v), k
    graph = orc.batch_egonets(shapes)
    assert np.array_equal(graph["src"].numpy(), z["src"]) and np.array_equal(graph["dst"].numpy(), z["dst"])
    assert np.array_equal(np.diff(graph["graph_off"].numpy()), z["batch_num_nodes"])
    return spec, z, shapes, x, q, params, graph


def oracle_masks(spec, params, graph):
    """explicit keep masks for the dropout cases, in the oracle's kwargs form."""
    if not spec.get("dropout"):
        return None
    pf, pa = spec["dropout"]
    is_gat = spec["prop"] in ("PGAT", "GAT")
    if is_gat:
        in_dims = [params[f"graph_propagate.gat_layers.{l}.fc.weight"].shape[1] for l in range(spec["num_layers"] + 1)]
    else:
        in_dims = [params[f"graph_propagate.layers.{l}.weight"].shape[0] for l in range(spec["num_layers"] + 1)]
    raw = gc.make_dropout_masks(spec, in_dims, graph["num_nodes"], int(graph["src"].numel()),
                                spec["heads"] if is_gat else None)
    out = []
    for fk, ak in raw:
        if is_gat:
            d = dict(feat_keep=torch.from_numpy(fk), feat_scale=1.0 / (1.0 - pf))
            if pa > 0:
                d.update(attn_keep=torch.from_numpy(ak), attn_scale=1.0 / (1.0 - pa))
            out.append(d)
        else:
            out.append(dict(keep=torch.from_numpy(fk), keep_scale=1.0 / (1.0 - pf)))
    return out


def check_grad(z, key, got, rtol, atol):
    got = np.asarray(got)
    if "grad:" + key in z:
        np.testing.assert_allclose(got, z["grad:" + key], rtol=rtol, atol=atol, err_msg=key)
    else:
        s, a, stride = z["gradstats:" + key]
        flat = got.reshape(-1)
        np.testing.assert_allclose(flat[::int(stride)], z["gradsample:" + key], rtol=rtol, atol=atol, err_msg=key)
        np.testing.assert_allclose(np.abs(flat.astype(np.float64)).sum(), a, rtol=1e-4, err_msg=key)


def oracle_gradients_f64(spec, params, graph, x, q, masks=None, with_x=False):
    """the oracle run in FLOAT64 on a golden case's inputs: {parameter name: gradient} (+ the gradient to the node features) -- the
    reference every gradient gate is measured against; the fp32 yardstick beside it is the unmodified reference's own golden gradient
    (or the same oracle in fp32)"""
    P = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in params.items()}
    xc = torch.from_numpy(x).double().requires_grad_(with_x)
    s, _hg, _hn = orc.taxoexpan_forward(P, graph, xc, torch.from_numpy(q).double(), spec["prop"], spec["readout"], spec["match"],
                                        spec["heads"], spec["num_layers"], masks)
    orc.info_nce_loss(s, spec["n_queries"]).backward()
    return {k: p.grad.numpy() for k, p in P.items()}, (xc.grad.numpy() if with_x else None)


def gradient_scale_floor(g64):
    """1e-3 of the largest gradient entry of the whole model: a tensor whose exact gradient is (nearly) zero -- attn_r of a one-head
    output layer: a constant added to every in-edge of a destination leaves its softmax unchanged -- is judged on that scale, not its own"""
    return 1e-3 * max(float(np.abs(np.asarray(v)).max()) for v in g64.values())


def gate_against_f64(got, ref64, yard, what, errors, report=None, factor=2.0, floor=2e-5, cap=1e-4, scale_floor=0.0):
    """the north star's "within 1e-4 fp32" for a gradient tensor: max |got - f64| <= factor x max |yardstick - f64| (the device is no
    further from the exact gradient than fp32 arithmetic in another summation order -- `yard` is the reference's own fp32 result or the
    fp32 oracle's), floored at `floor` x max |f64| (a tensor the yardstick happens to get to 1e-7 must not fail the device at 2e-7; both
    are maxima over thousands of entries, hence the factor 2); and never more than `cap` x max |f64| UNLESS fp32 arithmetic itself cannot
    reach that (PGCN's output bias gradient is a sum of 18 k cancelling terms: the fp32 oracle is 1.3e-4 off) -- then the factor decides"""
    got, ref64, yard = (np.asarray(a, dtype=np.float64) for a in (got, ref64, yard))
    assert got.shape == ref64.shape == yard.shape, (what, got.shape, ref64.shape, yard.shape)
    scale = max(float(np.abs(ref64).max()), scale_floor) or 1.0
    e_got, e_yard = float(np.abs(got - ref64).max()), float(np.abs(yard - ref64).max())
    if report is not None:
        report.append((what, e_got / scale, e_yard / scale))
    if not (e_got <= max(factor * e_yard, floor * scale) and e_got <= max(cap * scale, factor * e_yard)):
        errors.append(f"{what}: max |HIP - f64| = {e_got / scale:.3e} of max |ref|; the fp32 yardstick's is {e_yard / scale:.3e}")


def golden_grad_entries(z, key, *arrays):
    """the entries of a parameter gradient a golden file holds (all of them, or every stride-th): (golden values, the same entries of
    each of `arrays`)"""
    if "grad:" + key in z:
        return z["grad:" + key], [np.asarray(a) for a in arrays]
    stride = int(z["gradstats:" + key][2])
    return z["gradsample:" + key], [np.asarray(a).reshape(-1)[::stride] for a in arrays]

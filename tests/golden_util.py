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

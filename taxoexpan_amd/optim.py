"""`torch.optim.Adam` as the reference configures it (train.py:36 `config.initialize('optimizer', torch.optim, ...)`,
config.mag.json:66-73: Adam, lr 1e-3, weight_decay 0, amsgrad true), stepped by ONE HIP launch over all parameter tensors
(txe_adam_step).  Same constructor arguments, same `state_dict()` layout (step / exp_avg / exp_avg_sq / max_exp_avg_sq per
parameter), so optimizer checkpoints written by base_trainer.py:104-121 load into either class.  There is no CPU path."""
import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")        # the checks of torch.optim.Adam.__init__
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._tables = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)
        self._tables = {}

    def _init_state(self, p, amsgrad):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if amsgrad:
                st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _ptr_array(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ams = bool(group["amsgrad"])
            by_step = {}                                   # parameters that skipped updates (grad None) keep their own count
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("taxoexpan_amd.optim.Adam: dense fp32 parameters on the GPU only")
                st = self._init_state(p, ams)
                by_step.setdefault(int(st["step"]), []).append(p)
            for step, ps in by_step.items():
                key = (gi, tuple(id(p) for p in ps))
                tab = self._tables.get(key)
                if tab is None or any(t.data_ptr() != a for t, a in zip(tab["tensors"], tab["addr"])):
                    sts = [self.state[p] for p in ps]
                    for p, st in zip(ps, sts):
                        for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if ams else ()):
                            if not (st[k].is_contiguous() and st[k].dtype == torch.float32 and st[k].device == p.device):
                                st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
                    if any(not p.is_contiguous() for p in ps):
                        raise RuntimeError("taxoexpan_amd.optim.Adam: parameters must be contiguous")
                    tensors = list(ps) + [st["exp_avg"] for st in sts] + [st["exp_avg_sq"] for st in sts] + \
                        ([st["max_exp_avg_sq"] for st in sts] if ams else [])
                    tab = dict(tensors=tensors, addr=[t.data_ptr() for t in tensors], p=self._ptr_array(ps),
                               m=self._ptr_array([st["exp_avg"] for st in sts]), v=self._ptr_array([st["exp_avg_sq"] for st in sts]),
                               x=self._ptr_array([st["max_exp_avg_sq"] for st in sts]) if ams else None,
                               n=(C.c_longlong * len(ps))(*[p.numel() for p in ps]))
                    self._tables[key] = tab
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                _lib.call("txe_adam_step", len(ps), tab["p"], self._ptr_array(grads), tab["m"], tab["v"], tab["x"], tab["n"],
                          float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                          float(group["weight_decay"]), step + 1, _lib.stream_ptr())
                for p in ps:
                    self.state[p]["step"] += 1
        return loss

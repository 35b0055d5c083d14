"""``FusedAdamax`` - torch.optim.Adamax + torch.nn.utils.clip_grad_norm_ of the training step (Models/SDNetTrainer.py:307-317,
366-367) as three HIP launches over all trainable tensors (csrc/sdnet_optim.hip) instead of ~20 multi-tensor launches.

Same update rule, same defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay), same ``param_groups`` / ``zero_grad`` /
``state_dict`` surface, so it stands in for the ``'#'`` optimizer of the shipped configuration.  Extra: ``pinned`` maps an
embedding Parameter to the number of leading rows that are really trained; the remaining rows - which the trainer overwrites with
their fixed values after every step (:369-373) - are left out of the update (their gradients still enter the clipping norm,
exactly as in the reference)."""
import numpy as np
import torch

from . import hip

_CHUNK = 8192


class FusedAdamax:
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, pinned=None):
        self.params = [p for p in params]
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("FusedAdamax: contiguous fp32 device parameters only")
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps, "weight_decay": 0}]
        self.pinned = {id(p): int(n) for p, n in (pinned or {}).items()}
        self.state = {id(p): {"exp_avg": torch.zeros_like(p), "exp_inf": torch.zeros_like(p)} for p in self.params}
        self.step_count = 0
        self.steps = {id(p): 0 for p in self.params}      # torch.optim.Adamax counts every parameter's own steps
        self._plan_key = None
        self.norm_coef = None             # device tensor [total_norm, clip_coef] of the last clip_and_step

    # -- torch.optim surface --------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def state_dict(self):
        return {"step": self.step_count, "steps": [self.steps[id(p)] for p in self.params], "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "state": [{k: v.clone() for k, v in self.state[id(p)].items()} for p in self.params]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        for p, n in zip(self.params, sd.get("steps", [self.step_count] * len(self.params))):
            self.steps[id(p)] = int(n)
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        for p, st in zip(self.params, sd["state"]):
            for k in ("exp_avg", "exp_inf"):
                self.state[id(p)][k].copy_(st[k])

    # -- the step -----------------------------------------------------------------------------------------------------
    def _plan(self, live):
        """Chunk lists (norm: every element with a gradient; update: without the re-pinned rows) and the static pointer tables."""
        dev = live[0].device
        tn, sn, cn, tu, su, cu = [], [], [], [], [], []
        for t, p in enumerate(live):
            n = p.numel()
            n_upd = n
            if id(p) in self.pinned:
                n_upd = min(n, self.pinned[id(p)] * (n // p.shape[0]))
            for s in range(0, n, _CHUNK):
                tn.append(t); sn.append(s); cn.append(min(_CHUNK, n - s))
            for s in range(0, n_upd, _CHUNK):
                tu.append(t); su.append(s); cu.append(min(_CHUNK, n_upd - s))
        ints = torch.tensor(np.concatenate([tn, sn, cn, tu, su, cu]).astype(np.int32), device=dev)
        a, b = len(tn), len(tu)
        ptrs = torch.tensor([[p.data_ptr() for p in live], [self.state[id(p)]["exp_avg"].data_ptr() for p in live],
                             [self.state[id(p)]["exp_inf"].data_ptr() for p in live]], dtype=torch.int64, device=dev)
        self._plan_val = {"norm": (ints[0:a], ints[a:2 * a], ints[2 * a:3 * a], a),
                          "upd": (ints[3 * a:3 * a + b], ints[3 * a + b:3 * a + 2 * b], ints[3 * a + 2 * b:], b),
                          "ints": ints, "ptrs": ptrs, "partial": torch.empty(a, dtype=torch.float32, device=dev),
                          # gradient tensors are new every step: their pointer table is re-sent (two pinned staging buffers in
                          # turn, so the one of the previous step may still be in flight)
                          # per step and tensor: gradient pointer (int64) and lr / (1 - beta1^step) (fp32 bits in an int64 slot)
                          "gptr_host": [torch.empty(2 * len(live), dtype=torch.int64).pin_memory() for _ in range(2)],
                          "gptr": torch.empty(2 * len(live), dtype=torch.int64, device=dev)}
        self._plan_val["gptr_np"] = [h.numpy() for h in self._plan_val["gptr_host"]]
        self.norm_coef = torch.zeros(2, dtype=torch.float32, device=dev)

    def clip_and_step(self, max_norm=None, extra_sq=None):
        """clip_grad_norm_(params, max_norm) (skipped when None) followed by step().  The total norm and the clip coefficient
        stay on the device in ``self.norm_coef``.  ``extra_sq`` (device float tensor): the norm runs over the UPDATED elements only
        (the re-pinned embedding rows left out) and this squared norm is added in their place - dp.GradSync.pinned_sq."""
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        key = tuple(id(p) for p in live)
        if key != self._plan_key:
            self._plan(live)
            self._plan_key = key
        pl = self._plan_val
        slot = self.step_count & 1
        tab = pl["gptr_np"][slot]
        g0 = self.param_groups[0]
        n_live = len(live)
        clr = tab[n_live:].view(np.float32)            # first n_live float32 slots of the second half
        for i, p in enumerate(live):
            g = p.grad
            if not (g.is_contiguous() and g.dtype == torch.float32):
                g = p.grad = g.contiguous().float()
            tab[i] = g.data_ptr()
            self.steps[id(p)] += 1
            clr[i] = g0["lr"] / (1.0 - g0["betas"][0] ** self.steps[id(p)])
        pl["gptr"].copy_(pl["gptr_host"][slot], non_blocking=True)
        clr_dev = pl["gptr"][n_live:].view(torch.float32)
        lib = hip.load()
        st = hip.stream_ptr()
        self.step_count += 1
        coef = None
        if max_norm is not None:
            ct, cs, cc, n = pl["norm"] if extra_sq is None else pl["upd"]
            hip.check(lib.ruart_grad_norm_clip(hip.ptr(pl["gptr"]), hip.ptr(ct), hip.ptr(cs), hip.ptr(cc), n, float(max_norm),
                                               hip.ptr(pl["partial"]), hip.ptr(self.norm_coef), hip.ptr(extra_sq), st),
                      "ruart_grad_norm_clip")
            coef = self.norm_coef
        ct, cs, cc, n = pl["upd"]
        ptrs = pl["ptrs"]
        hip.check(lib.ruart_adamax_step(hip.ptr(ptrs[0]), hip.ptr(pl["gptr"]), hip.ptr(ptrs[1]), hip.ptr(ptrs[2]), hip.ptr(ct), hip.ptr(cs),
                                        hip.ptr(cc), n, hip.ptr(coef), hip.ptr(clr_dev), float(g0["betas"][0]), float(g0["betas"][1]),
                                        float(g0["eps"]), st), "ruart_adamax_step")
        # the kernel wrote the parameters through raw pointers: tell torch (version counters feed autograd's saved-tensor checks and
        # the trainable encoder's operand cache, bert_train16.accurate_weights)
        bump = getattr(torch.autograd.graph, "increment_version", None)
        if bump is None:
            raise RuntimeError("torch.autograd.graph.increment_version is missing: caches keyed on parameter versions "
                               "(bert_train16.accurate_weights) would go stale after this step")
        for p in live:
            bump(p)

    def step(self):
        self.clip_and_step(None)

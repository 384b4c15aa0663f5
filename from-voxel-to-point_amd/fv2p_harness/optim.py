"""AdamW for the benchmark's train step: torch.optim.AdamW(fused=True) with the per-step Python trimmed.

Same optimiser state, same two kernels per step (`torch._foreach_add_` on the step counters, `torch._fused_adamw_`),
hence bit-identical updates; what is skipped is torch.optim's per-step regrouping of the tensor lists by device and
dtype and its argument plumbing (~0.1 ms of host time per step here, where the training thread is launch bound).  The
reference trains with its own AdamW wrapper (tools/train_utils/optimization/: out of scope), so the optimiser is part of
the harness, not of the ops."""
import torch


class LeanAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._lean = None

    def _build(self):
        groups = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            st = [self.state[p] for p in ps]
            if any(len(s) == 0 for s in st) or group.get("amsgrad") or group.get("maximize") or group.get("capturable"):
                return None
            if any(p.device != ps[0].device or p.dtype != ps[0].dtype for p in ps):
                return None
            groups.append((group, ps, [s["exp_avg"] for s in st], [s["exp_avg_sq"] for s in st], [s["step"] for s in st],
                           sum(p.grad is None for p in group["params"])))
        return groups

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            return super().step(closure)
        if self._lean is None:
            out = super().step()          # first step: torch creates the state
            self._lean = self._build() or False
            return out
        if self._lean is False:
            return super().step()
        for group, ps, exp_avgs, exp_avg_sqs, steps, n_none in self._lean:
            grads = [p.grad for p in ps]
            if any(g is None for g in grads) or sum(p.grad is None for p in group["params"]) != n_none:
                self._lean = None         # the set of trained parameters changed: rebuild through torch's own step
                return super().step()
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(ps, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1, beta2=beta2,
                                weight_decay=group["weight_decay"], eps=group["eps"], maximize=False, grad_scale=None, found_inf=None)
        return None

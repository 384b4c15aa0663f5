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


@torch.no_grad()
def clip_grad_norm_(params, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=True) for parameters of one device and dtype (GRAD_NORM_CLIP,
    train_utils.py:43): the same kernels in the same order — `_foreach_norm`, the norm of the stacked norms, the clamped
    coefficient, `_foreach_mul_` — hence bit-identical gradients; skipped is the grouping of the tensors by device and dtype and
    the per-call checks (1.2 -> 0.3 ms of host time per step, at the step boundary where the device has nothing queued).
    Returns the total norm like the original."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    first = grads[0]
    if any(g.device != first.device or g.dtype != first.dtype for g in grads):
        return torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=True)
    norms = torch._foreach_norm(grads, 2.0)
    total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total

"""Optimisers behind the reference's OPTIMIZER_CLASSES interface (reference utils/optimization_utils.py:100-105).

`RAdam` keeps the reference's constructor, hyper-parameter groups, state layout (`step`, `exp_avg`, `exp_avg_sq` per
parameter, so optimizer.state_dict() round-trips with the reference's) and update rule (utils/optimization_utils.py:31-97),
but the update itself is ONE fused multi-tensor HIP launch sequence per (group, step count) instead of ~10 elementwise
kernels per parameter in a Python loop: the decoder has ~70 tensors / 2.85 M parameters, so a reference step is ~700 tiny
launches.  fp32 parameters on the GPU go through libqagnn_hip's qagnn_radam_step_f32; parameters that live on the CPU (unit
tests of the training driver) are updated with the same formulas in torch.
"""
import math

import torch
from torch.optim import SGD, Adam
from torch.optim.optimizer import Optimizer


def radam_step_size(step, beta1, beta2, degenerated_to_sgd=True):
    """(N_sma, step_size) of the rectified update at `step` (utils/optimization_utils.py:63-80)."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / \
            (1 - beta1 ** step)
    elif degenerated_to_sgd:
        step_size = 1.0 / (1 - beta1 ** step)
    else:
        step_size = -1
    return n_sma, step_size


class AdamW(Optimizer):
    """`transformers.AdamW` as the reference imports it (utils/optimization_utils.py:3; transformers == 3.4.0 is pinned in the
    reference's README, the class has since been removed from transformers): its defaults -- eps = 1e-6, weight_decay = 0.0,
    correct_bias = True -- AND its update rule, which is not torch.optim.AdamW's:

        m <- b1 m + (1 - b1) g;   v <- b2 v + (1 - b2) g^2
        p <- p - lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)        (eps is added to the UNcorrected sqrt(v): an effective
        p <- p - lr * weight_decay * p                                           eps of eps / sqrt(1 - b2^t), ~30x larger at t = 1;
                                                                                 the decay uses the parameter AFTER the Adam update)

    torch's class divides sqrt(v) by sqrt(1 - b2^t) before adding eps and applies the decay first.  The reference's driver sets
    weight_decay per parameter group and never passes eps (qagnn.py:196-206).  State layout as transformers': `step`, `exp_avg`,
    `exp_avg_sq` per parameter.  One multi-tensor (foreach) op sequence per group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            by_step = {}  # parameters that share a step count share their scalar factors
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                state['step'] += 1
                by_step.setdefault(state['step'], []).append(p)
            for t, ps in by_step.items():
                grads = [p.grad for p in ps]
                m = [self.state[p]['exp_avg'] for p in ps]
                v = [self.state[p]['exp_avg_sq'] for p in ps]
                torch._foreach_mul_(m, beta1)
                torch._foreach_add_(m, grads, alpha=1.0 - beta1)
                torch._foreach_mul_(v, beta2)
                torch._foreach_addcmul_(v, grads, grads, value=1.0 - beta2)
                denom = torch._foreach_sqrt(v)
                torch._foreach_add_(denom, group['eps'])
                step_size = group['lr']
                if group['correct_bias']:
                    step_size = step_size * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
                torch._foreach_addcdiv_(ps, m, denom, value=-step_size)
                if group['weight_decay'] > 0.0:
                    torch._foreach_mul_(ps, 1.0 - group['lr'] * group['weight_decay'])  # p <- p + (-lr wd) p, one rounding per element
        return loss


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.degenerated_to_sgd = degenerated_to_sgd
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            buckets = {}  # (step, on_gpu) -> lists: parameters that share a step count share step_size and the branch taken
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('RAdam does not support sparse gradients')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                state['step'] += 1
                fused = p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                buckets.setdefault((state['step'], fused, p.device if fused else None), []).append(p)
            for (step, fused, _dev), ps in buckets.items():
                n_sma, step_size = radam_step_size(step, beta1, beta2, self.degenerated_to_sgd)
                mode = 2 if n_sma >= 5 else (1 if step_size > 0 else 0)
                if fused:
                    from . import ops
                    grads = [p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous() for p in ps]
                    ops.kernels().radam_step(list(ps), grads, [self.state[p]['exp_avg'] for p in ps],
                                             [self.state[p]['exp_avg_sq'] for p in ps], beta1, beta2, group['eps'], group['lr'],
                                             group['weight_decay'], step_size, mode)
                    continue
                for p in ps:  # CPU tensors / other dtypes: the same update, tensor by tensor (utils/optimization_utils.py:52-93)
                    state = self.state[p]
                    grad = p.grad.float()
                    p32 = p.float()
                    m, v = state['exp_avg'], state['exp_avg_sq']
                    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                    m.mul_(beta1).add_(grad, alpha=1 - beta1)
                    if mode == 0:
                        continue
                    if group['weight_decay'] != 0:
                        p32.add_(p32, alpha=-group['weight_decay'] * group['lr'])
                    if mode == 2:
                        p32.addcdiv_(m, v.sqrt().add_(group['eps']), value=-step_size * group['lr'])
                    else:
                        p32.add_(m, alpha=-step_size * group['lr'])
                    p.copy_(p32)
        return loss


OPTIMIZER_CLASSES = {
    'sgd': SGD,
    'adam': Adam,
    'adamw': AdamW,   # transformers.AdamW's defaults (eps 1e-6, weight_decay 0) over torch.optim.AdamW's update
    'radam': RAdam,
}

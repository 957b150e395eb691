"""CPU restatement of the reference's RAdam update in float64 numpy.  TEST INFRASTRUCTURE ONLY.

Follows utils/optimization_utils.py:31-97 of the reference line by line (moments :57-58, the rectification term and step size
:60-80 incl. the degenerated_to_sgd branch, the two update branches :82-93, weight decay as `p += -wd * lr * p` :84-85, :90-91).
Pinned: tests/test_optimization.py holds it to tests/golden/radam.npz, which tests/golden/make_golden_radam.py produced by running
the reference's own RAdam class.  Only tests may import this module.
"""
import math

import numpy as np


def radam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, degenerated_to_sgd=True, dtype=np.float64):
    """One update of one tensor at (1-based) `step`; returns (p, m, v) as new arrays of `dtype`."""
    p, g, m, v = (np.asarray(x, dtype=dtype) for x in (p, g, m, v))
    v = v * dtype(beta2) + dtype(1 - beta2) * g * g            # :57  exp_avg_sq.mul_(beta2).addcmul_(1 - beta2, grad, grad)
    m = m * dtype(beta1) + dtype(1 - beta1) * g                # :58  exp_avg.mul_(beta1).add_(1 - beta1, grad)
    beta2_t = beta2 ** step                                    # :66
    n_sma_max = 2 / (1 - beta2) - 1                            # :67
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)     # :68
    if n_sma >= 5:                                             # :72-73
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / \
            (1 - beta1 ** step)
    elif degenerated_to_sgd:                                   # :74-75
        step_size = 1.0 / (1 - beta1 ** step)
    else:                                                      # :76-77
        step_size = -1
    if n_sma >= 5:                                             # :82-87
        if weight_decay != 0:
            p = p + dtype(-weight_decay * lr) * p
        p = p + dtype(-step_size * lr) * (m / (np.sqrt(v) + dtype(eps)))
    elif step_size > 0:                                        # :88-92
        if weight_decay != 0:
            p = p + dtype(-weight_decay * lr) * p
        p = p + dtype(-step_size * lr) * m
    return p, m, v


def transformers_adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0, correct_bias=True, dtype=np.float64):
    """One update of `transformers.AdamW` -- the class the reference imports for `--optim adamw` (utils/optimization_utils.py:3, :103).
    It is a THIRD-PARTY dependency that is not under /root/reference: transformers == 3.4.0 (pinned in the reference's README.md; the
    class was later deprecated and removed, it is not in the transformers of this image).  Restated from the published algorithm of
    that release (src/transformers/optimization.py, AdamW.step): first / second moments, eps added to the UNcorrected sqrt(v), the
    bias corrections folded into the step size, decoupled weight decay applied to the already-updated parameter.  Parity unpinned for
    this function (no golden vector of the third-party class exists in the reference); tests hold qagnn_amd's AdamW to it."""
    p, g, m, v = (np.asarray(x, dtype=dtype) for x in (p, g, m, v))
    m = m * dtype(beta1) + dtype(1.0 - beta1) * g
    v = v * dtype(beta2) + dtype(1.0 - beta2) * g * g
    denom = np.sqrt(v) + dtype(eps)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p + dtype(-step_size) * (m / denom)
    if weight_decay > 0.0:
        p = p + dtype(-lr * weight_decay) * p
    return p, m, v

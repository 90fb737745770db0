"""TEST INFRASTRUCTURE ONLY: numpy fp64 restatement of torch.optim.Adam's update (amsgrad=False, maximize=False,
weight_decay=0) with a per-row learning rate, as the reference drives it (/root/reference/scene/saro_gaussian.py:323,
:345-398).  Pinned in tests against torch.optim.Adam itself (scalar lr) -- torch is the reference's dependency here."""
import numpy as np


def step(p, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-15):
    """One step t (1-based).  lr: scalar or [rows]; arrays [rows, ...].  Returns (p, m, v) as fp64."""
    p, g, m, v = (np.asarray(a, np.float64) for a in (p, g, m, v))
    lr = np.asarray(lr, np.float64)
    if lr.ndim:
        lr = lr.reshape((-1,) + (1,) * (p.ndim - 1))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = np.sqrt(v) / np.sqrt(1 - b2 ** t) + eps
    p = p - (lr / (1 - b1 ** t)) * (m / denom)
    return p, m, v

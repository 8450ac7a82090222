"""Dense helpers shared by the tensorized code paths (reference: ``_legacy/utils.py:13-61``)."""

import torch


def scal(a, f, batch=False):
    """<a, f>: one dot product, or one per batch item (``utils.py:13-18``)."""
    if batch:
        B = a.shape[0]
        return (a.reshape(B, -1) * f.reshape(B, -1)).sum(1)
    return torch.dot(a.reshape(-1), f.reshape(-1))


def squared_distances(x, y):
    """Dense |x_i - y_j|^2 as |x|^2 - 2 x.y + |y|^2, for (N,D)x(M,D) or (B,N,D)x(B,M,D) (``utils.py:39-53``)."""
    if x.dim() not in (2, 3):
        print("x.shape : ", x.shape)
        raise ValueError("Incorrect number of dimensions")
    x2 = (x * x).sum(-1).unsqueeze(-1)  # (..., N, 1)
    y2 = (y * y).sum(-1).unsqueeze(-2)  # (..., 1, M)
    xy = torch.matmul(x, y.transpose(-1, -2))  # (..., N, M)
    return x2 - 2 * xy + y2


def distances(x, y):
    """Dense |x_i - y_j|, squared distances clamped at 1e-8 before the root (``utils.py:56-61``)."""
    return torch.sqrt(torch.clamp_min(squared_distances(x, y), 1e-8))

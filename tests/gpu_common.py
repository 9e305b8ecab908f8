"""helpers shared by the -m gpu tests"""
import numpy as np
import pytest


def get_canvas():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from canvas_amd import Canvas
    return Canvas(0)


def to_dev(a, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def pad16(a):
    n = (len(a) + 63) // 64 * 64
    out = np.zeros(n, a.dtype)
    out[:len(a)] = a
    return out

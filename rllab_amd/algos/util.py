"""Advantage helpers (mirror rllab/algos/util.py:7-12); numpy arrays or tensors."""
import numpy as np
import torch


def center_advantages(advantages):
    if torch.is_tensor(advantages):
        a = advantages.to(torch.float64)
        return (a - a.mean()) / (a.std(unbiased=False) + 1e-8)
    return (advantages - np.mean(advantages)) / (advantages.std() + 1e-8)


def shift_advantages_to_positive(advantages):
    if torch.is_tensor(advantages):
        return (advantages - advantages.min()) + 1e-8
    return (advantages - np.min(advantages)) + 1e-8

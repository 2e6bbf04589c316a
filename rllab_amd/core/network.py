"""MLP (API of rllab/core/network.py:36-101): a stack of dense layers whose parameters live in ONE
flat float32 device vector in the reference's order W0,b0,W1,b1,...,Wout,bout (W stored [in, out]
row-major), Glorot-uniform weights and zero biases like Lasagne's defaults.  The forward pass works on
"planes" (feature axis first, sample axis last), the engine's layout."""
import numpy as np
import torch

from rllab_amd.core.parameterized import Param


def rectify(x):
    return torch.relu(x)


tanh = torch.tanh


class MLP(object):
    def __init__(self, input_shape, output_dim, hidden_sizes, hidden_nonlinearity, output_nonlinearity=None,
                 name=None, offset=0):
        self.input_dim = int(np.prod(input_shape))
        self.output_dim = int(output_dim)
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        sizes = (self.input_dim,) + self.hidden_sizes + (self.output_dim,)
        prefix = (name + ".") if name else ""
        self.params, off = [], offset
        for li in range(len(sizes) - 1):
            lname = "output" if li == len(sizes) - 2 else "hidden_%d" % li
            w = Param("%s%s.W" % (prefix, lname), (sizes[li], sizes[li + 1]), off)
            off += w.size
            b = Param("%s%s.b" % (prefix, lname), (sizes[li + 1],), off, regularizable=False)
            off += b.size
            self.params += [w, b]
        self.end_offset = off

    def init_values(self, flat):
        """Write Glorot-uniform weights / zero biases into the numpy vector ``flat`` (np.random)."""
        for w in self.params[0::2]:
            bound = np.sqrt(6.0 / (w.shape[0] + w.shape[1]))
            flat[w.offset:w.offset + w.size] = np.random.uniform(-bound, bound, size=w.shape).reshape(-1)

    def forward_planes(self, x, flat):
        """x [Din, B] -> [Dout, B]."""
        h = x
        n_layers = len(self.params) // 2
        for li in range(n_layers):
            W = self.params[2 * li].view(flat)
            b = self.params[2 * li + 1].view(flat)
            h = W.t() @ h + b[:, None]
            if li < n_layers - 1:
                if self.hidden_nonlinearity is not None:
                    h = self.hidden_nonlinearity(h)
            elif self.output_nonlinearity is not None:
                h = self.output_nonlinearity(h)
        return h

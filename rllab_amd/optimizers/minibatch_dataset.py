"""Mini-batch iterator over an input tuple (API of rllab/optimizers/minibatch_dataset.py:4-38).

``BatchDataset(inputs, batch_size, extra_inputs)``: ``number_batches``, ``iterate(update=True)`` yields
``list(batch) + list(extra_inputs)`` for consecutive slices of a shuffled index, reshuffling (``update()``,
``np.random.shuffle``) after a full pass; ``batch_size=None`` yields the whole set once.

The reference slices numpy arrays along axis 0.  The engine's per-sample inputs are *planes* with the sample
axis LAST (obs [Do, B], advantages [B], ...) mixed with per-batch items (one log_std column, 1 / W): pass
``sample_axis=-1`` and the slice is taken on that axis of every tensor whose last dimension is the sample count,
everything else is handed through -- so the same class feeds ``FirstOrderOptimizer`` on device tensors.
"""
import numpy as np
import torch


class BatchDataset(object):
    def __init__(self, inputs, batch_size, extra_inputs=None, sample_axis=0):
        self._inputs = list(inputs)
        self._extra_inputs = list(extra_inputs) if extra_inputs is not None else []
        self._batch_size = batch_size
        self._axis = sample_axis
        self._n = self._inputs[0].shape[sample_axis]
        if batch_size is not None:
            self._ids = np.arange(self._n)
            self.update()

    @property
    def number_batches(self):
        if self._batch_size is None:
            return 1
        return int(np.ceil(self._n * 1.0 / self._batch_size))

    def _take(self, x, ids):
        per_sample = hasattr(x, "shape") and len(x.shape) > 0 and x.shape[self._axis] == self._n
        if not per_sample:
            return x
        if torch.is_tensor(x):
            return x.index_select(self._axis, torch.as_tensor(ids, device=x.device))
        return np.take(x, ids, axis=self._axis)

    def iterate(self, update=True):
        if self._batch_size is None:
            yield list(self._inputs) + list(self._extra_inputs)
            return
        for k in range(self.number_batches):
            ids = self._ids[k * self._batch_size:(k + 1) * self._batch_size]
            yield [self._take(x, ids) for x in self._inputs] + list(self._extra_inputs)
        if update:
            self.update()

    def update(self):
        np.random.shuffle(self._ids)

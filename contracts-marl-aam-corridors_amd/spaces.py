"""Minimal observation/action space objects.

The runner duck-types spaces by class NAME and a couple of attributes
(onpolicy/utils/util.py:32-52: `obs_space.__class__.__name__ == 'Box'` -> `.shape`;
`act_space.__class__.__name__ == 'Discrete'` -> `.n`; onpolicy/algorithms/utils/act.py:31-32),
so these two classes are drop-in for gym.spaces.Box / gym.spaces.Discrete on this path.
"""
import numpy as np


class Box(object):
    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)

    def __eq__(self, other):
        return other.__class__.__name__ == "Box" and tuple(other.shape) == self.shape


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return other.__class__.__name__ == "Discrete" and other.n == self.n

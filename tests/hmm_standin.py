"""Deterministic stand-in for ``hmmlearn.hmm.GaussianHMM`` -- TEST INFRASTRUCTURE.

hmmlearn is not installed in the build container or on the GPU box, and the HMM itself is outside the accelerated
path.  What IS on the drop-in boundary is everything the reference wraps around the model:
features in, the post-fit ``transmat_`` surgery (PlotEngine.py:422-438), supervised re-estimation
(PlotEngine.py:329-387), state -> event extraction (:447-475, :305-318), event merging.  To pin that logic this module
gives both sides -- the reference's PlotEngine when tests/golden/make_golden.py runs it, and this repository's
PlotEngine in the GPU tests -- the SAME small diagonal-Gaussian HMM with the attribute surface the reference touches
(``n_components, means_, covars_, transmat_, startprob_, fit, predict``).  It is deterministic (quantile
initialisation, hard EM), and ``predict`` is a Viterbi pass that really depends on ``transmat_`` / ``startprob_``, so
the surgery and the supervised parameters change its output the way they would change hmmlearn's.
"""
from __future__ import annotations

import numpy as np


class GaussianHMM:
    def __init__(self, n_components=1, covariance_type="diag", n_iter=10, random_state=None, **_ignored):
        self.n_components = int(n_components)
        self.covariance_type = covariance_type
        self.n_iter = int(n_iter)
        self.random_state = random_state

    # -- parameters -----------------------------------------------------------------------------------------------
    def _assign(self, x):
        d = ((x[:, None, :] - self.means_[None]) ** 2 / self.covars_[None]).sum(-1) + np.log(self.covars_).sum(-1)[None]
        return np.argmin(d, axis=1)

    def fit(self, x, lengths=None):
        x = np.asarray(x, np.float64)
        if x.ndim != 2 or len(x) < self.n_components:
            raise ValueError("not enough samples for the number of states")
        k, dim = self.n_components, x.shape[1]
        order = np.argsort(x[:, 0], kind="stable")
        parts = np.array_split(order, k)                         # quantile blocks of the first feature
        self.means_ = np.stack([x[p].mean(axis=0) for p in parts])
        self.covars_ = np.stack([x[p].var(axis=0) + 1e-3 for p in parts])
        states = self._assign(x)
        for _ in range(min(self.n_iter, 10)):
            for s in range(k):
                sel = states == s
                if sel.sum() >= 2:
                    self.means_[s] = x[sel].mean(axis=0)
                    self.covars_[s] = x[sel].var(axis=0) + 1e-3
            new = self._assign(x)
            if np.array_equal(new, states):
                break
            states = new
        trans = np.zeros((k, k))
        for a, b in zip(states[:-1], states[1:]):
            trans[a, b] += 1.0
        rows = trans.sum(axis=1, keepdims=True)
        self.transmat_ = np.where(rows > 0, trans / np.where(rows > 0, rows, 1.0), np.eye(k))
        self.startprob_ = np.full(k, 1.0 / k)
        assert dim == self.means_.shape[1]
        return self

    # -- decoding ---------------------------------------------------------------------------------------------------
    def predict(self, x, lengths=None):
        x = np.asarray(x, np.float64)
        means, var = np.asarray(self.means_, np.float64), np.asarray(self.covars_, np.float64)
        if var.ndim == 3:                                        # hmmlearn exposes diag covariances as full matrices
            var = np.stack([np.diag(v) for v in var])
        with np.errstate(divide="ignore"):
            log_t = np.log(np.asarray(self.transmat_, np.float64))
            log_s = np.log(np.asarray(self.startprob_, np.float64))
        ll = -0.5 * (((x[:, None, :] - means[None]) ** 2 / var[None]).sum(-1) + np.log(2 * np.pi * var).sum(-1)[None])
        n, k = ll.shape
        delta = log_s + ll[0]
        back = np.zeros((n, k), np.int64)
        for i in range(1, n):
            cand = delta[:, None] + log_t                        # [from, to]
            back[i] = np.argmax(cand, axis=0)
            delta = cand[back[i], np.arange(k)] + ll[i]
        path = np.empty(n, np.int64)
        path[-1] = int(np.argmax(delta))
        for i in range(n - 1, 0, -1):
            path[i - 1] = back[i, path[i]]
        return path


def install(sys_modules):
    """Register this class as ``hmmlearn.hmm.GaussianHMM`` in ``sys.modules`` (both sides import it from there)."""
    import types
    hm = types.ModuleType("hmmlearn.hmm")
    hm.GaussianHMM = GaussianHMM
    top = types.ModuleType("hmmlearn")
    top.hmm = hm
    sys_modules["hmmlearn"], sys_modules["hmmlearn.hmm"] = top, hm
    return hm

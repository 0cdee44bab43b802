"""anndata_standin.py - CONTAINER-ONLY stand-in for the `anndata` package (absent from this image), just large enough for
the reference's local-phasing path: xcltk/utils/csp_io.py:16-63 builds AnnData(X=None, obs=, var=) with dense layers and
transposes it; xcltk/baf/fc/main.py:113-131,419-454 reads .shape / .obs / .var, slices rows by a boolean Series, columns by
a boolean Series or by index labels, assigns a var column and copies.  Test infrastructure (golden generation) only."""
import numpy as np
import pandas as pd


class AnnData(object):
    def __init__(self, X=None, obs=None, var=None):
        self.obs = obs.copy().reset_index(drop=True)
        self.var = var.copy().reset_index(drop=True)
        self.obs.index = self.obs.index.astype(str)
        self.var.index = self.var.index.astype(str)
        self.layers, self.uns, self.obsm = {}, {}, {}
        if X is not None:                                     # (rdr/io.py:20-24 passes the count matrix as X; kept as one more layer)
            self.layers["X"] = np.asarray(X)

    @property
    def X(self):
        return self.layers.get("X")

    @property
    def shape(self):
        return (len(self.obs), len(self.var))

    def _new(self, obs, var, layers):
        a = AnnData.__new__(AnnData)
        a.obs, a.var, a.layers = obs, var, layers
        a.uns, a.obsm = dict(self.uns), {}
        return a

    def transpose(self):
        return self._new(self.var.copy(), self.obs.copy(), {k: np.asarray(v).T for k, v in self.layers.items()})

    def copy(self):
        return self._new(self.obs.copy(), self.var.copy(), {k: np.array(v) for k, v in self.layers.items()})

    @staticmethod
    def _positions(key, frame):
        if isinstance(key, slice):
            return np.arange(len(frame))[key]
        if isinstance(key, pd.Series):
            key = key.to_numpy()
        if isinstance(key, pd.Index):
            return frame.index.get_indexer(key)
        key = np.asarray(key)
        if key.dtype == bool:
            return np.flatnonzero(key)
        if key.dtype.kind in "OUS":
            return frame.index.get_indexer(key)
        return key.astype(np.int64)

    def __getitem__(self, key):
        r, c = key
        ri, ci = self._positions(r, self.obs), self._positions(c, self.var)
        return self._new(self.obs.iloc[ri].copy(), self.var.iloc[ci].copy(),
                         {k: np.asarray(v)[np.ix_(ri, ci)] for k, v in self.layers.items()})

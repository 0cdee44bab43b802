"""io.py - loading / saving the `basefc` output directory (features.tsv, barcodes.tsv, matrix.mtx).

Adapter of xcltk/rdr/io.py:14-74.  The reference returns an AnnData (cell x feature); `anndata` is an optional dependency
here: load_data() builds it when the package is importable, load_matrix_data() always works and returns the pieces
(features DataFrame, cells DataFrame, cell x feature scipy CSR matrix)."""
import os

import pandas as pd
from scipy import io as spio
from scipy import sparse


def load_features(fn):
    df = pd.read_csv(fn, header=None, sep="\t", dtype={0: str})
    df.columns = ["chrom", "start", "end", "feature"]
    return df


def load_cells(fn):
    df = pd.read_csv(fn, header=None)
    df.columns = ["cell"]
    return df


def load_matrix(fn, dense=True):
    m = spio.mmread(fn)
    return m.toarray() if dense else sparse.csr_matrix(m)


def save_matrix(mtx, fn):
    spio.mmwrite(fn, sparse.csr_matrix(mtx))


def save_features(df, fn):
    df.to_csv(fn, sep="\t", header=False, index=False)


def save_cells(df, fn):
    df.to_csv(fn, sep="\t", header=False, index=False)


def load_matrix_data(data_dir):
    """-> (features, cells, cell x feature CSR matrix); no anndata needed."""
    features = load_features(os.path.join(data_dir, "features.tsv"))
    cells = load_cells(os.path.join(data_dir, "barcodes.tsv"))
    return features, cells, load_matrix(os.path.join(data_dir, "matrix.mtx"), dense=False).T.tocsr()


def load_data(data_dir):
    """cell x feature AnnData as the reference builds it (rdr/io.py:14-26); needs the optional `anndata` package."""
    import anndata as ad
    features = load_features(os.path.join(data_dir, "features.tsv"))
    cells = load_cells(os.path.join(data_dir, "barcodes.tsv"))
    adata = ad.AnnData(X=load_matrix(os.path.join(data_dir, "matrix.mtx")), obs=features, var=cells)
    return adata.transpose()


def save_data(adata, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    save_cells(adata.obs, os.path.join(out_dir, "barcodes.tsv"))
    save_features(adata.var, os.path.join(out_dir, "features.tsv"))
    save_matrix(adata.X, os.path.join(out_dir, "matrix.mtx"))

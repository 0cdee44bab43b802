"""io.py - loading / saving the BAF feature-counting output directory (xcltk.region.tsv, xcltk.samples.tsv,
xcltk.{AD,DP,OTH}.mtx).

Adapter of xcltk/baf/io.py:14-81.  The reference returns an AnnData (cell x feature, layers AD / DP / OTH); `anndata` is an
optional dependency here: load_data() builds it when the package is importable, load_matrix_data() always works."""
import os

from scipy import sparse

from ..rdr.io import load_cells, load_features, load_matrix, save_cells, save_features, save_matrix

LAYERS = ("AD", "DP", "OTH")


def load_matrix_data(data_dir):
    """-> (features, cells, {"AD" | "DP" | "OTH": cell x feature CSR matrix}); no anndata needed."""
    features = load_features(os.path.join(data_dir, "xcltk.region.tsv"))
    cells = load_cells(os.path.join(data_dir, "xcltk.samples.tsv"))
    mats = {k: load_matrix(os.path.join(data_dir, "xcltk.%s.mtx" % k), dense=False).T.tocsr() for k in LAYERS}
    return features, cells, mats


def load_data(data_dir):
    """cell x feature AnnData with layers AD / DP / OTH (baf/io.py:14-33); needs the optional `anndata` package."""
    import anndata as ad
    features = load_features(os.path.join(data_dir, "xcltk.region.tsv"))
    cells = load_cells(os.path.join(data_dir, "xcltk.samples.tsv"))
    adata = ad.AnnData(X=None, obs=features, var=cells)
    for k in LAYERS:
        adata.layers[k] = load_matrix(os.path.join(data_dir, "xcltk.%s.mtx" % k))
    return adata.transpose()


def save_data(adata, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    save_cells(adata.obs, os.path.join(out_dir, "xcltk.samples.tsv"))
    save_features(adata.var, os.path.join(out_dir, "xcltk.region.tsv"))
    for k in LAYERS:
        save_matrix(sparse.csr_matrix(adata.layers[k]), os.path.join(out_dir, "xcltk.%s.mtx" % k))

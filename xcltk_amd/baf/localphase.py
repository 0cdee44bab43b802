"""localphase.py - local phasing of the SNPs of one block (host-side float64, not on the GPU).

Restates xcltk/baf/localphase.py:14-343 (the reference took `Local_Phasing` from XClone): all SNPs of a block are assumed to
share one allelic ratio per cell, so deciding which SNPs have their haplotypes swapped is a two-component clustering of the
SNPs, fitted by EM; neighbouring SNPs are pulled together by a Gaussian kernel over their coordinates; cells whose
aggregated B-allele frequency is near 0.5 carry no phase information and are dropped, round after round.

The results that matter downstream are DISCRETE (flip / no flip per SNP), but they are thresholds of float64 quantities.
To make the same decisions as the reference on the same inputs the arithmetic below keeps the reference's operation order
(same reductions over the same axes, same clipping constants); only the structure of the code is this repo's own.
tests/golden/phasing holds flip vectors produced by the reference's own functions.
"""
import numpy as np
from scipy.special import logsumexp

EPS_THETA = 1e-6                  # localphase.py:173: thetas are clipped into [eps, 1 - eps] before the logs


def gaussian_smooth(v, x, b, a=0):
    """u[i] = sum_j v[j] w_ij / sum_j w_ij with w_ij = exp(a - (x_j - x_i)^2 / b^2)   (localphase.py:245-270)."""
    v = np.asarray(v)
    u = v.copy()
    for i in range(len(v)):
        w = np.exp(a - (x - x[i]) ** 2 / b ** 2)
        w = w / np.sum(w)
        u[i] = np.sum(v * w)
    return u


def _loglik(AD, BD, thetas):
    t = thetas.copy()
    t[t <= 0] = EPS_THETA
    t[t >= 1] = 1 - EPS_THETA
    return AD @ np.log(t) + BD @ np.log(1 - t)


def em_two_haplotypes(AD, DP, positions, min_iter=10, max_iter=1000, tol=1e-3, kernel_b=20000):
    """EM of localphase.py:139-242 (`Local_Phasing`, warm start, Gaussian-kernel smoothing of the assignments).
    AD, DP: SNP x cell counts.  Returns (Z, thetas, loglik): Z[:, 0] = P(SNP keeps its phase)."""
    N = AD.shape[0]
    BD = DP - AD
    Z = np.zeros((N, 2))
    Z[:, 0] = (AD.sum(1) / DP.sum(1)).reshape(-1)
    Z[:, 1] = 1 - Z[:, 0]
    depth = DP.T.sum(1, keepdims=True)

    def m_step(Z):
        return np.array((AD.T @ Z + BD.T @ (1 - Z)) / depth)
    thetas = m_step(Z)
    mat = _loglik(AD, BD, thetas)
    ll_new = np.sum(logsumexp(mat, axis=1))
    for it in range(max_iter):
        ll_old = ll_new + 0.0
        E = np.exp(np.array(mat) - np.max(np.array(mat), axis=-1, keepdims=True))
        Z = E / np.sum(E, axis=-1, keepdims=True)
        Z[:, 0] = gaussian_smooth(Z[:, 0], positions, b=kernel_b)
        Z[:, 1] = 1 - Z[:, 0]
        thetas = m_step(Z)
        mat = _loglik(AD, BD, thetas)
        ll_new = np.sum(logsumexp(mat, axis=1))
        if it >= min_iter and ll_new - ll_old < tol:
            break
    return Z, thetas, ll_new


def snp_local_phasing(AD, DP, positions, min_iter=5, max_iter=50, min_expr_snps=1, low_baf=0.45, up_baf=0.55):
    """Iterative phasing of localphase.py:14-135.  AD, DP: cell x SNP counts with AD counted on the CURRENT reference
    haplotype.  Returns the boolean flip vector, or None when no informative cell is left."""
    with np.errstate(all="ignore"):
        keep = DP.sum(axis=1) > 0
        AD, DP = AD[keep, :], DP[keep, :]
        if AD.shape[0] <= 0:
            return None
        keep = (DP > 0).sum(axis=1) >= min_expr_snps
        AD, DP = AD[keep, :], DP[keep, :]
        if AD.shape[0] <= 0:
            return None
        flip_final = None
        for i in range(max_iter):
            baf = AD.sum(axis=1) / DP.sum(axis=1)
            keep = np.logical_or(baf < low_baf, baf > up_baf)
            AD, DP = AD[keep, :], DP[keep, :]
            if AD.shape[0] <= 0:
                return None
            Z, _, _ = em_two_haplotypes(AD.T, DP.T, positions)
            flip = np.array(Z[:, 1] >= Z[:, 0])
            flip_final = flip if i == 0 else np.logical_xor(flip_final, flip)
            if i > 0 and i >= min_iter and (np.all(flip) or np.all(np.logical_not(flip))):
                return flip_final
            AD = AD * (1 - flip.T) + (DP - AD) * flip.T
        return flip_final

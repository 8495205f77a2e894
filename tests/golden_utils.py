"""Helpers around the reference's stored outputs (tests/golden/reference_outdata.npz).

See tests/golden/make_golden.py for provenance and for why a fixed point of the VB loop can be
replayed from the stored posterior alone when the model is linear in its parameters.
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_reference_outdata():
    return np.load(os.path.join(GOLDEN_DIR, "reference_outdata.npz"))


def unpack(mvn_rows, n):
    """[rows][V] packed MVN -> cov [V][n][n], means [V][n] (dist_mvn.cc:347-374)."""
    V = mvn_rows.shape[1]
    cov = np.zeros((V, n, n))
    k = 0
    for r in range(n):
        for c in range(r + 1):
            cov[:, r, c] = cov[:, c, r] = mvn_rows[k]
            k += 1
    means = np.array(mvn_rows[k:k + n].T, dtype=np.float64)
    assert np.all(mvn_rows[k + n] == 1.0)
    return cov, means


def poly_design(T, degree):
    t = np.arange(1, T + 1, dtype=np.float64)
    return np.stack([t ** n for n in range(degree + 1)], axis=1)


def data_with_same_sufficient_statistics(J, cov, means, prior_prec=1e-12, b0=1e6, seed=0):
    """Rebuild, for every voxel, a length-T series whose J'y and y'y reproduce the stored fixed
    point (posterior cov/means of the P model parameters + the noise gamma).

    cov [V][P+1][P+1], means [V][P+1] as unpacked from the reference's finalMVN.
    Returns float64 [T][V].
    """
    T, P = J.shape
    V = cov.shape[0]
    G = J.T @ J
    rng = np.random.default_rng(seed)
    # Orthonormal basis of the complement of col(J) - any unit residual direction will do.
    Q, _ = np.linalg.qr(J)
    y = np.zeros((T, V))
    for v in range(V):
        Sigma = cov[v, :P, :P]
        m = means[v, :P]
        phi_mean = means[v, P]
        phi_var = cov[v, P, P]
        b = phi_var / phi_mean  # GammaDist::SetMeanVariance, dist_gamma.cc:29-33
        Lam = prior_prec * np.eye(P) + phi_mean * G
        a = np.linalg.solve(G, Lam @ m) / phi_mean
        kk = 2.0 * (1.0 / b - 1.0 / b0) - np.trace(Sigma @ G)
        kk -= (a - m) @ G @ (a - m)
        assert kk > 0
        r = rng.standard_normal(T)
        r -= Q @ (Q.T @ r)
        r -= Q @ (Q.T @ r)
        r *= np.sqrt(kk) / np.linalg.norm(r)
        y[:, v] = J @ a + r
    return y

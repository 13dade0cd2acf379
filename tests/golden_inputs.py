"""Seeded synthetic inputs shared by tests/golden/make_golden.py and the tests.

np.random.RandomState is numpy's frozen legacy generator, so these streams are reproducible
across numpy versions; every fixture also stores a SHA-1 of the inputs it was generated from
and the tests compare it before trusting the fixture.

Shapes follow SURVEY.md section 8(d).
"""
import numpy as np


def gmm_unit(n, d, n_comp, seed, dtype, sigma=0.35, nonneg=False):
    """n unit-norm d-vectors from an n_comp-component isotropic Gaussian mixture (so that
    nearest neighbours are meaningful); nonneg=True mimics post-ReLU CNN features."""
    rs = np.random.RandomState(seed)
    centers = rs.randn(n_comp, d)
    comp = rs.randint(0, n_comp, size=n)
    x = centers[comp] + sigma * rs.randn(n, d)
    if nonneg:
        np.maximum(x, 0.0, out=x)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return x.astype(dtype)


def perturbed_queries(X, nq, seed, eps=0.05):
    rs = np.random.RandomState(seed)
    idx = rs.choice(X.shape[0], nq, replace=False)
    q = X[idx].astype(np.float64) + eps * rs.randn(nq, X.shape[1]) / np.sqrt(X.shape[1])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(X.dtype), idx


def c1_inputs():
    rs = np.random.RandomState(1234)
    X = rs.randn(100000, 128).astype(np.float32)
    Q = rs.randn(100, 128).astype(np.float32)
    return X, Q


def c2_inputs():
    X = gmm_unit(50000, 128, 256, 2, np.float64)
    Q, _ = perturbed_queries(X, 64, 22)
    return X, Q


def descriptor_params(d=128, n_comp=4096, seed=9):
    """Parameters of the descriptor-like distribution of the bench workload (config C4): an anisotropic mixture
    whose spectrum decays as j^-1/2 in a random orthonormal basis, with far more (mild) components than the
    codebooks can memorise, so that the residuals are continuous and LOPQ codes are almost all distinct -- as for
    real CNN / face descriptors.  (The 256-component isotropic mixture of c2 gives 84 % duplicate codes.)"""
    rs = np.random.RandomState(seed)
    basis, _ = np.linalg.qr(rs.randn(d, d))
    scale = (1.0 + np.arange(d)) ** -0.5
    centers = (rs.randn(n_comp, d) * scale).dot(basis.T)
    mean = (0.5 * rs.randn(d) * scale).dot(basis.T)
    return basis, scale, centers, mean


def descriptor_like(n, d, seed, dtype=np.float64):
    basis, scale, centers, mean = descriptor_params(d)
    rs = np.random.RandomState(seed)
    comp = rs.randint(0, centers.shape[0], size=n)
    x = mean + 0.7 * centers[comp] + 0.7 * (rs.randn(n, d) * scale).dot(basis.T)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return x.astype(dtype)


def c4_inputs():
    X = descriptor_like(100000, 128, 4, np.float64)
    Q, _ = perturbed_queries(X, 64, 44)
    return X, Q


def c3_inputs():
    X = gmm_unit(20000, 320, 128, 3, np.float32, nonneg=True)
    Q, _ = perturbed_queries(X, 32, 33)
    return X, Q


def c3b_inputs():
    X = gmm_unit(8000, 288, 64, 5, np.float32, nonneg=True)
    Q, _ = perturbed_queries(X, 16, 55)
    return X, Q


def c3full_inputs():
    """C3 at its true input width: 6000 x 4096 float32 non-negative unit vectors (post-ReLU-like), 32 queries."""
    X = gmm_unit(6000, 4096, 64, 6, np.float32, nonneg=True)
    Q, _ = perturbed_queries(X, 32, 66)
    return X, Q


def c3full_train():
    """Training set of the c3full model: 20000 vectors of the same mixture (round 3: the first fit used the 6000 index
    vectors, which left 21 of the 32 local rotations with fewer points than dimensions and the index degenerate)."""
    return gmm_unit(20000, 4096, 64, 6, np.float32, nonneg=True)


def pk_inputs():
    """Inputs of the pickled-model fixtures (tests/golden/pk): 3000 x 24 float64, 8 queries."""
    rs = np.random.RandomState(5150)
    centers = rs.randn(40, 24)
    X = centers[rs.randint(0, 40, size=3000)] + 0.5 * rs.randn(3000, 24)
    Q = X[rs.choice(3000, 8, replace=False)] + 0.05 * rs.randn(8, 24)
    return X, Q


def tiny_inputs():
    rs = np.random.RandomState(7)
    X = rs.randn(4000, 8)  # float64 -> float64 coarse centroids
    Q = rs.randn(12, 8)
    return X, Q

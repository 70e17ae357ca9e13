"""Harmonic analysis on the 60-element icosahedral rotation group for the group convolution.

The group conv of the reference, out_o(g) = sum_c sum_{h in H} W[o,c,h] x_c(h g) (13-point stencil H, network/group_feat.py:20-33),
is a correlation on the group, so it block-diagonalises over the five real irreducible representations rho of dimensions
d = 1,3,3,4,5 (sum d^2 = 60).  With the orthonormal transform F[(rho,i,j), g] = sqrt(d/60) rho(g)_ij (Peter-Weyl):

    out~_o(rho) = sum_c  W^_{oc}(rho) . x~_c(rho)        ([d x d] . [d x d]),   W^_{oc}(rho) = sum_k W[o,c,k] rho(h_k)^T

i.e. per (o,c) pair sum_d d^3 = 244 multiply-adds instead of 60*13 = 780: the same result with 3.2x fewer MACs, as five
dense GEMMs.  The transforms themselves are per-channel 60x60 products (cheap next to the channel-mixing GEMMs) and the
pointwise BatchNorm/ReLU stay in the group domain.

Irreps are obtained numerically from the multiplication table alone: a random symmetric operator commuting with the left
regular representation has the irreducible subspaces as eigenspaces; one eigenspace per isomorphism type (told apart by
its character) defines rho(g) = V^T L_g V to machine precision (the stored 3x3 rotation table only closes to 6e-5).
"""
import functools

import numpy as np

from .group import tables, G

DIMS = (1, 3, 3, 4, 5)


class GroupFourier:
    def __init__(self, T=None, seed=1):
        T = T or tables()
        P = T.P                                   # P[a,g] = index(R_g R_a)
        self.P = P
        # left-regular action used by the conv: (S_h x)(g) = x(h g) = x[P[g,h]]  (Nei[g,k] = P[g,H[k]])
        S = np.zeros((G, G, G))
        for h in range(G):
            S[h, np.arange(G), P[:, h]] = 1.0      # (S_h x)[g] = x[P[g,h]]
        self.S = S
        # an operator commuting with every S_h: right translations x(g) -> x(g a), i.e. x[P[a,g]]
        rng = np.random.default_rng(seed)
        w = rng.standard_normal(G)
        A = np.zeros((G, G))
        for a in range(G):
            Ra = np.zeros((G, G)); Ra[np.arange(G), P[a, :]] = 1.0
            A += w[a] * Ra
        for h in range(G):
            assert np.abs(A @ S[h] - S[h] @ A).max() < 1e-9
        A = A + A.T
        evals, evecs = np.linalg.eigh(A)
        # group eigenvalues into eigenspaces
        spaces = []
        i = 0
        while i < G:
            j = i + 1
            while j < G and abs(evals[j] - evals[i]) < 1e-7:
                j += 1
            spaces.append(evecs[:, i:j]); i = j
        dims = sorted(v.shape[1] for v in spaces)
        assert dims == sorted([1] + [3] * 6 + [4] * 4 + [5] * 5), dims
        chars = [np.array([np.trace(V.T @ S[h] @ V) for h in range(G)]) for V in spaces]
        reps = []
        for d in (1, 4, 5):
            V = next(V for V in spaces if V.shape[1] == d)
            reps.append(V)
        threes = [(V, c) for V, c in zip(spaces, chars) if V.shape[1] == 3]
        c0 = threes[0][1]
        other = next(V for V, c in threes if np.abs(c - c0).max() > 1e-6)
        reps = [reps[0], threes[0][0], other, reps[1], reps[2]]
        # rho_r(h) = V^T S_h V ; S is an anti-homomorphism or homomorphism depending on conventions -- only the identities
        # verified in tests (orthogonality of F, the convolution theorem) are relied upon.
        self.rho = [np.stack([V.T @ S[h] @ V for h in range(G)]) for V in reps]       # [60,d,d] each
        for r, d in zip(self.rho, DIMS):
            assert r.shape == (G, d, d)
            assert np.abs(np.einsum('hij,hkj->hik', r, r) - np.eye(d)).max() < 1e-10
        # orthonormal transform: rows q = (rho, i, j), columns g
        rows = []
        self.index = []                                                                # q -> (rho index, i, j)
        for ri, (r, d) in enumerate(zip(self.rho, DIMS)):
            for i in range(d):
                for j in range(d):
                    rows.append(np.sqrt(d / 60.0) * r[:, i, j]); self.index.append((ri, i, j))
        self.F = np.stack(rows)                                                        # [60 (q), 60 (g)]
        assert np.abs(self.F @ self.F.T - np.eye(G)).max() < 1e-10
        self.offsets = np.cumsum([0] + [d * d for d in DIMS])
        self._find_convention(T)

    def _find_convention(self, T):
        """Determine, by brute force on a random function, how the stencil shift x -> x[Nei[:,k]] acts on the coefficient
        matrices: x~'(rho) = M . x~(rho) (left) or x~(rho) . M (right), with M = rho(h_k) or its transpose."""
        rng = np.random.default_rng(0)
        x = rng.standard_normal(G)
        k = 5
        h = int(T.H[k])
        xs = x[T.Nei[:, k]]
        found = None
        for side in ('left', 'right'):
            for tr in (False, True):
                ok = True
                for ri, d in enumerate(DIMS):
                    X = self.coef_matrix(self.F @ x, ri); Xs = self.coef_matrix(self.F @ xs, ri)
                    M = self.rho[ri][h].T if tr else self.rho[ri][h]
                    pred = M @ X if side == 'left' else X @ M
                    ok &= np.abs(pred - Xs).max() < 1e-9
                if ok:
                    found = (side, tr)
        assert found is not None, 'no shift convention matches'
        self.side, self.transpose = found

    def coef_matrix(self, coef, ri):
        d = DIMS[ri]
        return coef[self.offsets[ri]:self.offsets[ri + 1]].reshape(d, d)

    def stencil_matrices(self, T=None):
        """M[ri][k] (d x d): the action of stencil position k on irrep ri's coefficient matrix (see _find_convention)."""
        T = T or tables()
        return [np.stack([(self.rho[ri][int(h)].T if self.transpose else self.rho[ri][int(h)]) for h in T.H]) for ri in range(5)]

    def transform_weights(self, W):
        """W [O,C,13] -> per irrep W^ [O, C, d, d] with out~_o = sum_c W^_{oc} . x~_c ('left') or x~_c . W^_{oc} ('right')."""
        Ms = self.stencil_matrices()
        return [np.einsum('ock,kij->ocij', W.astype(np.float64), M) for M in Ms]


@functools.lru_cache(maxsize=None)
def group_fourier():
    return GroupFourier()

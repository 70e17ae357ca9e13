"""The 60-element icosahedral rotation group shared by every kernel on the path.

Reference: utils/group_related/{Rotation,60_60,Nei_Index_in_SO3_ordered_13}.npy, loaded at
network/group_feat.py:12-14, network/rot_detect.py:39-40, network/eqv_trans.py:83-86,
network/rot_coh_match.py:127,330, test/estimator.py:78,374-375.

Only the rotations R (in the reference's order) and the 13-element conv stencil H are stored
(roreg_amd/data/icosahedral_group.npz, built by tools/make_group_tables.py which asserts that the
derived tables equal the reference's files).  Derived here:

    P[a, g]   = index(R_g . R_a)        -- "60_60.npy"; x[..., P[a]] rotates a descriptor by R_a
    Nei[g, k] = P[g, H[k]]             -- "Nei_Index_in_SO3_ordered_13.npy"; Nei[g,0] == g
"""
import os
import functools
import numpy as np

_DATA = os.path.join(os.path.dirname(__file__), 'data', 'icosahedral_group.npz')

G = 60      # group order
K = 13      # stencil size


class GroupTables:
    def __init__(self, R, H):
        R = np.asarray(R, np.float64)
        H = np.asarray(H, np.int64)
        assert R.shape == (G, 3, 3) and H.shape == (K,)
        # prod[a,g] = R_g . R_a ; nearest group element (the stored table closes to ~6e-5)
        prod = np.einsum('gij,ajk->agik', R, R)
        d = np.abs(prod[:, :, None] - R[None, None]).reshape(G, G, G, 9).max(-1)
        P = d.argmin(-1).astype(np.int64)
        assert d.min(-1).max() < 1e-3
        for a in range(G):
            assert np.array_equal(np.sort(P[a]), np.arange(G)) and np.array_equal(np.sort(P[:, a]), np.arange(G))
        self.R = R
        self.H = H
        self.P = P
        self.Nei = np.ascontiguousarray(P[:, H])
        assert np.array_equal(self.Nei[:, 0], np.arange(G))
        # inverse element index: R_inv[a] with R_{inv[a]} = R_a^T
        dinv = np.abs(np.transpose(R, (0, 2, 1))[:, None] - R[None]).reshape(G, G, 9).max(-1)
        self.inv = dinv.argmin(-1).astype(np.int64)

    # -- helpers used by the ET path: which group columns of each layer can reach output g=0 -------
    def live_sets(self, depth):
        """live[d] = set of group indices whose value at conv depth d (counted from the output)
        can influence column g=0 after d stencil hops.  live[0]={0}, live[1]=Nei[0], ..."""
        cur = {0}
        out = [sorted(cur)]
        for _ in range(depth):
            nxt = set()
            for g in cur:
                nxt.update(int(v) for v in self.Nei[g])
            cur = nxt
            out.append(sorted(cur))
        return out

    def export_reference_files(self, directory):
        """Write the three .npy files in the reference's on-disk format (float64 containers)."""
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, 'Rotation.npy'), self.R)
        np.save(os.path.join(directory, '60_60.npy'), self.P.astype(np.float64))
        np.save(os.path.join(directory, 'Nei_Index_in_SO3_ordered_13.npy'), self.Nei.astype(np.float64))


@functools.lru_cache(maxsize=None)
def tables(directory=None):
    """Group tables.  If `directory` (cfg.SO3_related_files) holds the reference's three files they
    are used (and cross-checked against the built-in ones); otherwise the built-in tables."""
    z = np.load(_DATA)
    builtin = GroupTables(z['R'], z['H'])
    if directory and os.path.exists(os.path.join(directory, 'Rotation.npy')):
        R = np.load(os.path.join(directory, 'Rotation.npy'))
        Nei = np.load(os.path.join(directory, 'Nei_Index_in_SO3_ordered_13.npy')).astype(np.int64)
        P = np.load(os.path.join(directory, '60_60.npy')).astype(np.int64)
        t = GroupTables(R, Nei[0])
        if not (np.array_equal(t.P, P) and np.array_equal(t.Nei, Nei)):
            raise ValueError(f'group tables under {directory} are not self-consistent')
        return t
    return builtin

"""Build roreg_amd/data/icosahedral_group.npz from the reference's group tables.

Runs in the build container only (reads /root/reference, which does not exist on the GPU box).
The three reference files (utils/group_related/{Rotation,60_60,Nei_Index_in_SO3_ordered_13}.npy)
are *data* that fix the ordering of the 60 icosahedral rotations; trained checkpoints depend on
that ordering, so it cannot be regenerated independently.  We store only what is irreducible:
  R    (60,3,3) float64  the rotations, in the reference's order
  H    (13,)    int64    the conv stencil  (= Nei[0])
and re-derive the permutation table P and the neighbour table Nei from R, H, asserting equality with
the reference's files:
  P[a,g]   = index(R_g . R_a)
  Nei[g,k] = P[g,H[k]] = index(R_{H[k]} . R_g)
"""
import numpy as np, os, sys

REF = '/root/reference/utils/group_related'
OUT = os.path.join(os.path.dirname(__file__), '..', 'roreg_amd', 'data', 'icosahedral_group.npz')


def index_of(M, R):
    d = np.abs(R - M[None]).reshape(60, -1).max(1)
    i = int(np.argmin(d))
    assert d[i] < 1e-3 and np.sort(d)[1] > 0.1  # table closes to ~6e-5 only
    return i


def main():
    R = np.load(f'{REF}/Rotation.npy')
    P_ref = np.load(f'{REF}/60_60.npy').astype(np.int64)
    N_ref = np.load(f'{REF}/Nei_Index_in_SO3_ordered_13.npy').astype(np.int64)
    assert R.shape == (60, 3, 3) and np.allclose(np.linalg.det(R), 1) and np.allclose(R[0], np.eye(3))
    P = np.zeros((60, 60), np.int64)
    for a in range(60):
        for g in range(60):
            P[a, g] = index_of(R[g] @ R[a], R)
    assert (P == P_ref).all(), 'P[a,g] = index(R_g R_a) does not reproduce 60_60.npy'
    H = N_ref[0].copy()
    Nei = P[:, H]
    assert (Nei == N_ref).all(), 'Nei[g,k] = P[g,H[k]] does not reproduce Nei_Index_in_SO3_ordered_13.npy'
    assert (Nei[:, 0] == np.arange(60)).all()
    np.savez(OUT, R=R, H=H)
    print('wrote', OUT, 'H =', H.tolist())


if __name__ == '__main__':
    main()

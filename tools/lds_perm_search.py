"""Search for the LDS slot order of the ET trunk convolution's activation slab (csrc/group_conv.hip, group_conv_split_kernel).

The kernel's B-operand reads are gathered: lane n of a 16-lane ds_read_b128 group reads the 16-byte k-octet of input column
gather[j(n)][k] of keypoint kp(n) at slot  kp * S + sigma[column]  (16 slots = the 64 banks).  With the natural order (sigma = identity,
S = 48) the 13 gathered columns of a keypoint collide modulo 16: on average 3.2 lanes of a group want the same bank quad, i.e. 65-69 % of
the kernel's LDS cycles are bank conflicts (profiles/r02_bench_pmc: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.65).  This tool searches
(simulated annealing over injective maps of the 45 live columns into S slots) for the order that minimises the mean serialisation over
all 13 stencil positions and all tile alignments, and prints it as the table roreg_amd/network/eqv_trans.py carries.

    python tools/lds_perm_search.py [S=45] [iterations=60000] [restarts=4]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from roreg_amd.group import tables


def read_groups(gather, Lout, KS, ncol_tile=128):
    """(keypoint offset, input column) of the 16 lanes of every b128 read group: all stencil positions, all tile alignments."""
    kp, col = [], []
    for col0 in range(0, Lout * ncol_tile, ncol_tile):                  # the alignments repeat with period lcm(Lout, 128)
        b_first = col0 // Lout
        for blk in range(ncol_tile // 16):
            lanes = col0 + blk * 16 + np.arange(16)
            for k in range(KS):
                kp.append(lanes // Lout - b_first); col.append(gather[lanes % Lout, k])
    return np.array(kp), np.array(col)


def serialisation(sigma, S, KP, COL):
    q = (KP * S + sigma[COL]) % 16
    return (q[:, :, None] == np.arange(16)[None, None, :]).sum(1).max(1).mean()


def et_trunk_gather():
    T = tables()
    live = T.live_sets(2)
    pos2 = {g: i for i, g in enumerate(live[2])}
    return np.array([[pos2[int(v)] for v in T.Nei[g]] for g in live[1]])      # [13 output columns, 13 stencil positions] -> 45 live inputs


if __name__ == '__main__':
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    restarts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    gb = et_trunk_gather()
    KP, COL = read_groups(gb, 13, 13)
    print('identity, S = 48 (round 2): %.3f;  identity, S = %d: %.3f' % (serialisation(np.arange(45), 48, KP, COL), S, serialisation(np.arange(45), S, KP, COL)))
    rng = np.random.default_rng(0)
    best = None
    for r in range(restarts):
        sigma = rng.permutation(S)[:45]
        free = [s for s in range(S) if s not in set(sigma.tolist())]
        c = serialisation(sigma, S, KP, COL)
        temp = 0.03
        for it in range(iters):
            new = sigma.copy()
            a = int(rng.integers(0, 45)); fi = None
            if free and rng.random() < 0.3:
                fi = int(rng.integers(0, len(free))); old = int(new[a]); new[a] = free[fi]
            else:
                b = int(rng.integers(0, 45)); new[a], new[b] = new[b], new[a]
            cn = serialisation(new, S, KP, COL)
            if cn <= c or rng.random() < np.exp((c - cn) / temp):
                if fi is not None:
                    free[fi] = old
                sigma, c = new, cn
            temp = max(0.0015, temp * 0.9998)
        print(f'restart {r}: {c:.4f}', flush=True)
        if best is None or c < best[0]:
            best = (c, sigma.copy())
    print('best mean serialisation %.4f with S = %d' % (best[0], S))
    print('sigma =', best[1].tolist())

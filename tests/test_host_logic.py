"""Host-side logic that needs no GPU: the C-ABI surface, config defaults, dataset parsing, sharding + gloo gather."""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def test_cabi_exports_every_declared_symbol():
    """libroreg_hip.so loads on a CPU-only box and exports exactly what include/roreg_hip.h declares."""
    from roreg_amd import hip
    header = open(os.path.join(ROOT, 'include', 'roreg_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = sorted(set(re.findall(r'\b(roreg_\w+)\s*\(', header)))
    assert len(declared) >= 18
    L = hip.lib()
    for name in declared:
        assert hasattr(L, name), f'{name} declared in roreg_hip.h but not exported'
        assert name in hip.PROTOTYPES, f'{name} has no ctypes prototype'
    assert sorted(hip.PROTOTYPES) == declared
    assert L.roreg_abi_version() == hip.ABI_VERSION == 6
    assert int(re.search(r'#define\s+ROREG_ABI_VERSION\s+(\d+)', header).group(1)) == hip.ABI_VERSION
    # pure host entry points work without a GPU
    assert L.roreg_group_conv_packed_size(256, 512, 13) == 13 * 256 * 512
    assert L.roreg_group_conv_packed_size(64, 16, 13) == 13 * 64 * 32          # Cout padded to 32


def test_weight_packing_layout():
    from roreg_amd import hip
    rng = np.random.default_rng(0)
    Cout, Cin, KS = 48, 16, 13
    W = rng.standard_normal((Cout, Cin, KS)).astype(np.float32)
    n = hip.lib().roreg_group_conv_packed_size(Cin, Cout, KS)
    out = np.full(n, np.nan, np.float32)
    assert hip.lib().roreg_group_conv_pack_weights(W.ctypes.data, Cin, Cout, KS, out.ctypes.data) == 0
    P = out.reshape(KS, Cin // 8, 64, 2, 4)                     # [k][c/8][o (padded)][h][r]
    for (k, cb, o, h, r) in [(0, 0, 0, 0, 0), (12, 1, 47, 1, 3), (5, 0, 17, 1, 2)]:
        assert P[k, cb, o, h, r] == W[o, cb * 8 + 2 * r + h, k]
    assert (P[:, :, 48:] == 0).all()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from roreg_amd import hip
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.setattr(hip, '_LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(hip.HipError):
        hip.lib()


def test_round5_entry_points_validate_before_they_touch_the_gpu():
    """The v5 additions refuse bad arguments with a message (no HIP call has happened: this runs on a CPU-only box), and the two run-time
    switches report and restore their setting."""
    from roreg_amd import hip
    L = hip.lib()
    assert L.roreg_linear_cat3(None, None, None, None, 10, 16, None, None, 64, None, None) != 0
    assert b'roreg_linear_cat3' in L.roreg_last_error()
    assert L.roreg_ft_nonlin_packed(None, None, None, None, None, None, 60, 60, 0, 32, None, None, 0, None) != 0
    assert b'roreg_ft_nonlin_packed' in L.roreg_last_error()
    for switch in (L.roreg_gemm_persistent, L.roreg_linear_path):
        was = switch(-1)
        assert was in (0, 1) and switch(1 - was) == was and switch(-1) == 1 - was
        assert switch(was) == 1 - was and switch(-1) == was
        assert switch(7) == was and switch(-1) == was                          # anything else is a query
    assert L.roreg_gemm_persistent(2) == 0 and L.roreg_gemm_persistent(0) == 2 and L.roreg_gemm_persistent(-1) == 0      # (the GEMM has a third form)


def test_host_tensor_is_rejected_not_computed_on_cpu():
    import torch
    from roreg_amd import hip
    with pytest.raises(hip.HipError):
        hip._ptr(torch.zeros(4))


def test_parses_defaults_match_reference():
    from roreg_amd.parses.parses_test import default_config
    c = default_config()
    assert (c.GF, c.RD, c.RM, c.ET, c.testset, c.keynum, c.max_iter) == ('yoho_des', False, False, 'yohoc', '3dmatch', 5000, 1000)
    assert (c.ransac_ird, c.tau_1, c.tau_2, c.tau_3, c.match_n, c.bs_GF, c.bs_ET) == (0.1, 0.05, 0.1, 0.2, 0.5, 1250, 1000)
    assert c.output_cache_fn == './data/YOHO_FCGF/Testset' and c.model_fn == './checkpoints/FCGF' and c.backbone == 'FCGF'
    assert c.SO3_related_files == './utils/group_related'


def test_state_dict_keys_match_shipped_checkpoints():
    """RD / RM mirrors load the reference's shipped weights strict=True (key names and shapes)."""
    import torch
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    for kind, f in [('RD_test', 'weights_RD'), ('RM_test', 'weights_RM')]:
        net = name2network[kind](default_config())
        sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f + '.npz')).items()}
        net.load_state_dict(sd, strict=True)
    gf = name2network['GF_test'](default_config()).state_dict()
    assert tuple(gf['PartI_net.Conv_in.0.weight'].shape) == (256, 32, 1, 13)
    assert tuple(gf['PartI_net.SO3_Conv_layers.0.comb_layer_in.2.weight'].shape) == (512, 256, 1, 13)
    et = name2network['ET_test'](default_config()).state_dict()
    assert tuple(et['PartII_To_R_FC.6.weight'].shape) == (4, 128, 1, 1) and 'Conv_init.comb_layer.0.running_mean' in et


def test_gt_log_parser_on_reference_demo_file(tmp_path):
    from roreg_amd.dataops.dataset import ThrDMatchPartDataset
    root = tmp_path / 'kitchen'
    (root / 'PointCloud').mkdir(parents=True)
    (root / 'PointCloud' / 'gt.log').write_bytes(open(os.path.join(GOLDEN, 'demo_gt.log'), 'rb').read())
    ds = ThrDMatchPartDataset(str(root), 2)
    assert ds.pair_ids == [('0', '1')] and ds.pc_ids == ['0', '1']
    T = ds.get_transform('0', '1')
    assert T.dtype == np.float32 and T.shape == (3, 4)
    assert np.allclose(T, [[0.141, 0.989, 0.034, -2.247], [-0.903, 0.114, 0.414, -1.131], [0.405, -0.089, 0.910, 0.673]])


def test_ply_reader_and_keypoints(tmp_path):
    from roreg_amd.dataops.dataset import ThrDMatchPartDataset, read_ply_points
    rng = np.random.default_rng(1)
    pts = rng.uniform(-2, 2, (50, 3)).astype(np.float32)
    root = tmp_path / 's'
    (root / 'PointCloud').mkdir(parents=True); (root / 'Keypoints').mkdir()
    hdr = 'ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nend_header\n'
    rec = np.zeros(50, dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('red', 'u1')])
    rec['x'], rec['y'], rec['z'] = pts[:, 0], pts[:, 1], pts[:, 2]
    (root / 'PointCloud' / 'cloud_bin_0.ply').write_bytes(hdr.encode() + rec.tobytes())
    asc = 'ply\nformat ascii 1.0\nelement vertex 50\nproperty double x\nproperty double y\nproperty double z\nend_header\n' + \
          '\n'.join(' '.join(repr(float(v)) for v in p) for p in pts) + '\n'
    (root / 'PointCloud' / 'cloud_bin_1.ply').write_text(asc)
    (root / 'PointCloud' / 'gt.log').write_text('0\t1\t2\n1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n')
    assert np.array_equal(read_ply_points(str(root / 'PointCloud' / 'cloud_bin_0.ply')), pts.astype(np.float64))
    assert np.array_equal(read_ply_points(str(root / 'PointCloud' / 'cloud_bin_1.ply')), pts.astype(np.float64))
    idx = np.array([3, 7, 49, 0])
    np.savetxt(root / 'Keypoints' / 'cloud_bin_0Keypoints.txt', idx)
    ds = ThrDMatchPartDataset(str(root), 2)
    k = ds.get_kps('0')
    assert k.dtype == np.float64 and np.array_equal(k, pts[idx].astype(np.float64))
    assert np.array_equal(np.load(root / 'Keypoints_PC' / 'cloud_bin_0Keypoints.npy'), k)
    assert np.array_equal(ds.get_kps('0'), k)           # served from the cache


def test_dropin_aliases():
    import importlib
    from roreg_amd import dropin
    saved = {k: sys.modules.get(k) for k in dropin._ALIASES}
    try:
        dropin.install()
        import network, test, utils.knn_search, parses.parses_test     # noqa: F401
        assert set(test.name2estimator) == {'yohoc', 'yohoo'} and set(test.name2matcher) == {'matmul', 'yoho_mat'}
        assert set(network.name2network) >= {'GF_test', 'RD_test', 'RM_test', 'ET_test'}
        assert utils.knn_search.knn_module.KNN(5).k == 5
        from test.evaluator import yoho_evaluator                       # noqa: F401
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_yohoc_sampling_replays_the_reference_generator_calls():
    """hip.yohoc_draw (a HOST function of the C-ABI) against the literal loop of test/estimator.py:220-230 on the global generator:
    same bins, same triples, same generator state afterwards; the vectorised bin statistics and the stacked 3-point Kabsch against
    their per-item forms, bitwise."""
    from roreg_amd import hip
    from oracle import ref_numpy as O
    from roreg_amd.test.estimator import yohoc_ransac, dr_bins, three_point_transforms

    class Cfg:
        ransac_ird = 0.1
    ref = yohoc_ransac(Cfg())
    rng = np.random.default_rng(0)
    for trial in range(12):
        M = int(rng.integers(5, 4000))
        dr = rng.integers(0, 60, M) if trial % 3 else rng.choice([3, 7, 7, 7, 11], M)
        if trial == 5:
            dr = np.arange(M) % 60
        if trial == 7:
            dr = np.arange(60)[:min(M, 60)]                              # every bin has < 2 members
        stat, prob = O.dr_statistic(dr)                                  # the oracle's literal loop (pinned to the reference goldens)
        counts, members, starts, prob_v = dr_bins(dr)
        stat_api, prob_api = ref.DR_statictic(dr)                        # the stage class's method is built on dr_bins
        assert np.array_equal(prob_api, prob) and (stat is None) == (stat_api is None) and (stat is None or stat_api == stat)
        assert np.array_equal(prob, prob_v)
        if stat is None:
            assert np.sum(prob_v) < 1e-5
            continue
        for r in range(60):
            assert np.array_equal(members[starts[r]:starts[r] + counts[r]], np.array(stat[r], np.int64))
        max_iter = int(rng.choice([1, 10, 1000]))
        np.random.seed(trial)
        want_bins, want_idx, tries = [], [], 0
        while len(want_bins) < max_iter:
            if tries > 50000:
                break
            tries += 1
            R = np.random.choice(range(60), p=prob)
            if len(stat[R]) < 2:
                continue
            want_idx.append(np.random.choice(np.array(stat[R]), 3)); want_bins.append(R)
        tail = np.random.rand(3)
        np.random.seed(trial)
        bins, picks = hip.yohoc_draw(prob_v, counts, max_iter)
        assert np.array_equal(tail, np.random.rand(3))                      # generator advanced by exactly the same amount
        assert np.array_equal(bins, np.array(want_bins))
        assert np.array_equal(members[starts[bins][:, None] + picks], np.array(want_idx).reshape(-1, 3))
        # the engine's per-pair streams: a private RandomState(seed) gives the draws np.random.seed(seed) gives, ends in the same state,
        # and leaves the process-global generator untouched
        np.random.seed(4242); before = np.random.get_state()[1].copy()
        own = np.random.RandomState(trial)
        bins_p, picks_p = hip.yohoc_draw(prob_v, counts, max_iter, rng=own)
        assert np.array_equal(bins_p, bins) and np.array_equal(picks_p, picks) and np.array_equal(own.rand(3), tail)
        assert np.array_equal(np.random.get_state()[1], before)
    # stacked Kabsch == per-triple Kabsch (same LAPACK call per matrix), including repeated points
    K0 = rng.uniform(0, 3, (500, 3)); K1 = K0[:, [1, 2, 0]] + rng.normal(0, 0.01, (500, 3))
    idx = rng.integers(0, 500, (300, 3)); idx[::7, 1] = idx[::7, 0]; idx[::13] = idx[::13, :1]
    got = three_point_transforms(K0[idx], K1[idx])
    for i in range(idx.shape[0]):
        k0, k1 = K0[idx[i]], K1[idx[i]]
        c0 = np.mean(k0, 0, keepdims=True); c1 = np.mean(k1, 0, keepdims=True)
        U, S, VT = np.linalg.svd((k1 - c1).T @ (k0 - c0))
        R = VT.T @ U.T
        assert np.array_equal(got[i], np.concatenate([R, (c0 - c1 @ R.T).T], 1))


def test_shard_scenes_balanced_and_complete():
    from roreg_amd.distributed import shard_scenes
    counts = dict(zip('abcdefgh', [449, 217, 159, 207, 104, 54, 292, 138]))       # 3DMatch-like pair counts
    for world in [1, 2, 4, 8]:
        sh = shard_scenes(counts, world)
        assert len(sh) == world
        seen = {}
        for r in sh:
            for scene, a, b in r:
                assert 0 <= a < b <= counts[scene]
                seen.setdefault(scene, []).append((a, b))
        for scene, n in counts.items():
            rs = sorted(seen[scene])
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(len(rs) - 1))
        loads = [sum(b - a for _, a, b in r) for r in sh]
        assert max(loads) <= 1.25 * sum(counts.values()) / world + 1


def test_mt_shuffle_prefix_replays_numpys_seeded_shuffles():
    """roreg_mt_shuffle_prefix (host C, threaded) == np.random.seed(seed); shuffle(arange(n))[:take] for consecutive lists on one stream:
    the matcher's keypoint sampling (two lists per pair) and the one-shot estimator's hypothesis order (one list per pair)."""
    from roreg_amd import hip
    rng = np.random.default_rng(5)
    seeds = np.concatenate([[0, 1, 2 ** 32 - 1, 2 ** 32 + 5, 123456789], rng.integers(0, 2 ** 40, 60)])
    for per_job, take, hi in ((2, 500, 5200), (1, 1000, 4000), (3, 7, 1)):
        sizes = rng.integers(0, hi + 1, (seeds.shape[0], per_job))
        sizes[0, 0] = 0; sizes[1, 0] = 1; sizes[2, 0] = 2
        for nt in (1, 5):
            got = hip.mt_shuffle_prefix(seeds, sizes, take, n_threads=nt)
            for j, seed in enumerate(seeds):
                st = np.random.RandomState(int(seed) % (2 ** 32))
                for l in range(per_job):
                    x = np.arange(int(sizes[j, l])); st.shuffle(x)
                    want = np.full(take, -1, np.int64); want[:min(take, x.shape[0])] = x[:take]
                    assert np.array_equal(got[j, l], want), (per_job, take, j, l, int(sizes[j, l]))
    # and the process-global spelling gives the same stream as RandomState(seed)
    np.random.seed(77); a = np.arange(300); np.random.shuffle(a)
    assert np.array_equal(hip.mt_shuffle_prefix([77], [[300]], 300)[0, 0], a)


def test_global_stream_shuffles_leave_numpys_generator_where_numpy_would():
    """roreg_mt_stream_shuffle_prefix behind hip.global_stream_shuffle_prefix: the shuffles of an unseeded Test.py (test/matcher.py:83-88,
    test/estimator.py:423-425) drawn from the process-global generator in C -- the same lists as np.random.shuffle, and the generator's state
    afterwards is numpy's (whatever is drawn next agrees), from any position of the 624-word block, with a cached gaussian kept."""
    from roreg_amd import hip
    rng = np.random.default_rng(11)
    for trial, (warm, take) in enumerate(((0, 500), (3, 1000), (623, 7), (624, 5000), (1251, 64))):
        sizes = rng.integers(0, 5200, 23); sizes[:3] = (0, 1, 2)
        np.random.seed(1000 + trial)
        np.random.random_sample(warm)                               # (position inside the block)
        if trial == 2:
            np.random.standard_normal()                             # (leaves a cached gaussian in the legacy state)
        start = np.random.get_state()
        want = []
        for n in sizes:
            x = np.arange(int(n)); np.random.shuffle(x)
            w = np.full(take, -1, np.int64); w[:min(take, x.shape[0])] = x[:take]; want.append(w)
        after = (np.random.randint(0, 2 ** 31, 9), np.random.standard_normal(3), np.random.rand(2))
        np.random.set_state(start)
        got = hip.global_stream_shuffle_prefix(sizes, take)
        assert np.array_equal(got, np.stack(want))
        assert np.array_equal(np.random.randint(0, 2 ** 31, 9), after[0]) and np.array_equal(np.random.standard_normal(3), after[1]) and np.array_equal(np.random.rand(2), after[2])
    assert hip.global_stream_shuffle_prefix([], 10).shape == (0, 10)


def test_scene_pair_lists_touch_every_cloud_and_are_reproducible():
    """bench.py's synthetic pair lists (both kinds): the requested number of distinct pairs (i < j), sorted, every cloud touched (the chain
    (i, i+1) is always in), the same list for the same seed; with locality the pairs sit closer to the diagonal than uniformly drawn ones."""
    from roreg_amd import synth
    for n_clouds, n_pairs in ((60, 449), (37, 54), (16, 60)):
        spans = {}
        for loc in (None, 8.0):
            a = synth.scene_pair_list(n_clouds, n_pairs, 901, locality=loc)
            assert a == synth.scene_pair_list(n_clouds, n_pairs, 901, locality=loc)
            assert len(a) == n_pairs and len(set(a)) == n_pairs and a == sorted(a)
            assert all(0 <= i < j < n_clouds for i, j in a)
            assert {i for p in a for i in p} == set(range(n_clouds))
            spans[loc] = np.mean([j - i for i, j in a])
        if n_pairs > 2 * n_clouds:
            assert spans[8.0] < spans[None]


def test_shard_scenes_full_benchmark_shape_with_pair_lists():
    """bench.py's plan: exact per-range cloud counts; complete at every world size; modelled efficiency (one-rank cost / (N x makespan)) >= 0.97
    at 2 and 4 ranks and >= 0.83 at 8 with uniformly random pair lists, >= 0.88 with bench.py's default lists (scan-sequence-like locality):
    a scene's slices each re-extract the clouds they touch; the wrap-around fill cuts at most one scene per rank boundary."""
    from roreg_amd import synth
    from roreg_amd.distributed import shard_scenes
    names = synth.THREEDMATCH_SCENES
    clouds = dict(zip(names, synth.THREEDMATCH_CLOUDS)); npairs = dict(zip(names, synth.THREEDMATCH_PAIRS))
    CC = 9.0                                                          # the planner's cloud cost (distributed.shard_scenes default)
    cost1 = sum(npairs.values()) + CC * sum(clouds.values())
    for world, floor, locality in ((1, 0.999, None), (2, 0.97, None), (4, 0.97, None), (8, 0.83, None), (8, 0.88, 8.0)):
        lists = {s: synth.scene_pair_list(clouds[s], npairs[s], 900 + i, locality=locality) for i, s in enumerate(names)}
        plan = shard_scenes(npairs, world, clouds, pair_lists=lists)
        seen = {}
        loads = []
        for r in plan:
            load = 0.0
            for scene, a, b in r:
                assert 0 <= a < b <= npairs[scene]
                seen.setdefault(scene, []).append((a, b))
                load += (b - a) + CC * len({i for pr in lists[scene][a:b] for i in pr})
            loads.append(load)
        for scene, n in npairs.items():
            rs = sorted(seen[scene])
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(len(rs) - 1))
        assert cost1 / (world * max(loads)) >= floor, (world, cost1 / (world * max(loads)))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from roreg_amd.distributed import shard_scenes, gather_table, ROW
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
dist.init_process_group('gloo', rank=rank, world_size=world)
counts = {'s0': 7, 's1': 3, 's2': 5}
mine = shard_scenes(counts, world)[rank]
rows = []
for scene, a, b in mine:
    for i in range(a, b):
        r = np.zeros(ROW); r[0] = int(scene[1:]); r[1] = i; r[2] = i + 1; r[3] = 100 + i; r[5:20] = np.arange(15) + rank; r[20] = 0.5
        rows.append(r)
table = gather_table(np.array(rows).reshape(-1, ROW))
assert table.shape == (15, ROW), table.shape
keys = sorted((int(r[0]), int(r[1])) for r in table)
assert keys == sorted((s, i) for s, n in [(0, 7), (1, 3), (2, 5)] for i in range(n)), keys
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


def test_gather_table_world_size_2_gloo(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = str(s.getsockname()[1]); s.close()
    w = tmp_path / 'worker.py'
    w.write_text(_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(w), ROOT, str(r), '2', port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def run_distributed_worlds(tmp_path, mode, timeout=600):
    """run_distributed.evaluate(seed=...) at world size 1 and 2 (tests/_dist_worker.py) -> the two rank-0 result files."""
    worker = os.path.join(ROOT, 'tests', '_dist_worker.py')
    env = dict(os.environ, OMP_NUM_THREADS='1')
    outs = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = str(s.getsockname()[1]); s.close()
        work = tmp_path / f'w{world}'; work.mkdir()
        out = str(tmp_path / f'result_w{world}.npz')
        procs = [subprocess.Popen([sys.executable, worker, ROOT, str(work), str(r), str(world), port, mode, out], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, env=env) for r in range(world)]
        logs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
        for p, o in zip(procs, logs):
            assert p.returncode == 0 and 'ok' in o, o
        outs[world] = np.load(out)
    return outs[1], outs[2]


def test_evaluate_does_not_depend_on_the_number_of_ranks(tmp_path):
    """--seed: every pair draws from its own generator stream, so the result table at world size 2 (one scene cut across the ranks by
    the shard plan, the whole multi-rank control flow over gloo) equals the one at world size 1.  Host logic only (stub engine); the
    same check with the real engine is a GPU test (tests/test_hip_pipeline.py)."""
    one, two = run_distributed_worlds(tmp_path, 'stub')
    assert int(two['split']) == 1                                # the plan really cut a scene
    assert sorted(one.files) == sorted(two.files)
    for k in one.files:
        if k != 'split':
            assert np.array_equal(one[k], two[k]), k


def test_batch_create_matches_reference():
    """extractor_localtrans.batch_create (test/estimator.py:293-306): slices, float32 / int64 casts and the 0 <-> 1 exchange."""
    from roreg_amd.test.estimator import extractor_localtrans
    z = np.load(os.path.join(GOLDEN, 'batch_create.npz'))
    ex = extractor_localtrans.__new__(extractor_localtrans)
    b = ex.batch_create(z['f0_fcgf'], z['f1_fcgf'], z['f0_yomo'], z['f1_yomo'], z['index_pre'], 2, 7)
    assert sorted(b) == ['after_eqv0', 'after_eqv1', 'before_eqv0', 'before_eqv1', 'pre_idx']
    for k, v in b.items():
        assert str(v.dtype) == ('torch.int64' if k == 'pre_idx' else 'torch.float32') and np.array_equal(v.numpy(), z[k]), k


def run_forced_collectives(backend, out, timeout=300, extra=()):
    """tests/_nccl_worker.py in a fresh process -> (return code, output)."""
    worker = os.path.join(ROOT, 'tests', '_nccl_worker.py')
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.pop('MASTER_PORT', None)
    p = subprocess.run([sys.executable, worker, ROOT, backend, str(out), *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=timeout)
    return p.returncode, p.stdout.decode()


def test_forced_collectives_on_one_rank_gloo(tmp_path):
    """ROREG_FORCE_COLLECTIVES: a one-rank group still runs the table's all_gather (with and without the plan's row counts) and drives
    run_plan's whole exchange branch with transfers from the rank to itself; the results are bitwise those of the plain pass.  Host
    control flow over gloo with the stub engine; the same worker runs the RCCL path on the GPU (tests/test_hip_pipeline.py)."""
    rc, log = run_forced_collectives('gloo', tmp_path / 'forced.npz')
    assert rc == 0 and 'ok' in log, log
    z = np.load(tmp_path / 'forced.npz')
    assert z['table'].shape == (37, 21) and z['rows'].shape[1] == 20


def test_watchdog_ends_a_hung_process_with_status_1(tmp_path):
    """distributed.watchdog: a block that does not return in time ends the process (status 1, stacks on stderr) -- and leaves a block that
    does return alone, nested blocks included."""
    rc, log = run_forced_collectives('gloo', tmp_path / 'x.npz', extra=('hang',), timeout=60)
    assert rc == 1 and 'Timeout' in log and 'did not fire' not in log, log
    from roreg_amd.distributed import watchdog
    import time
    with watchdog(5.0, 'outer'):
        with watchdog(0.5, 'inner'):
            pass
        time.sleep(0.8)                                      # the inner block's timer must be gone; the outer one re-armed with its own deadline
    with watchdog(0.0, 'disabled'):
        pass
    time.sleep(0.2)


def _bench_line(argv, env_extra, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS='1', ROREG_BENCH_ENGINE='tests._bench_stub:make', PYTHONPATH=ROOT, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), p.stdout.decode()[-2000:]       # stdout carries the record and NOTHING else
    return json.loads(lines[0])


def test_bench_launcher_starts_n_ranks_and_gathers_one_table():
    """`python bench.py --gpus 3` with no torch.distributed environment must itself start 3 ranks (the driver's command line), print ONE
    JSON line with n_gpus = 3, shard the pairs, ship the cut scene's extractor outputs between the ranks and gather one result table.
    Host logic only: the engine is the stand-in of tests/_bench_stub.py over gloo.  The accuracy block (FMR / IR / RR over all ranks) and
    the result table must not depend on the number of ranks nor on whether the exchange is used."""
    args = ['--steps', '1', '--warmup', '1', '--kpts', '24', '--all-steps', '1']
    one = _bench_line(args + ['--gpus', '1'], {})
    assert one['n_gpus'] == 1 and one['config']['eqv_transfers_per_step'] == 0
    two = _bench_line(args + ['--gpus', '3'], {})                            # (3 ranks: the smallest world size whose plan cuts a scene)
    assert two['n_gpus'] == 3 and two['scaling'] == 'strong' and two['config']['pairs_per_step'] == 1623
    assert two['config']['eqv_transfers_per_step'] > 0                       # a scene was cut and its clouds travelled
    assert sum(two['config']['cloud_extractions_per_rank']) == 433           # every cloud extracted exactly once
    assert len(two['config']['shard_plan']) == 3 and two['config']['backend'] == 'gloo'
    assert two['accuracy'] == one['accuracy'] and one['accuracy']['pairs'] == 1623
    plain = _bench_line(args + ['--gpus', '3', '--no-exchange'], {})
    assert plain['config']['eqv_transfers_per_step'] == 0 and sum(plain['config']['cloud_extractions_per_rank']) > 433
    assert plain['accuracy'] == one['accuracy']
    eight = _bench_line(args + ['--gpus', '8'], {})                          # the driver's largest launch: every scene cut, 85 transfers
    assert eight['n_gpus'] == 8 and sum(eight['config']['cloud_extractions_per_rank']) == 433 and eight['accuracy'] == one['accuracy']
    assert eight['config']['eqv_bytes_moved_per_step'] == eight['config']['eqv_transfers_per_step'] * 24 * 32 * 60 * 4
    assert eight['config']['result_table_bytes_gathered_per_step'] == 1623 * 21 * 8


def test_bench_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, ROREG_BENCH_ENGINE='tests._bench_stub:make', PYTHONPATH=ROOT, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--kpts', '24'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode != 0 and b'--gpus 2' in p.stderr


def test_exchange_plan_extracts_every_cloud_once():
    """Cut scenes: with the extractor-output exchange the ranks of an 8-rank plan together extract exactly the benchmark's 433 clouds
    (the reference extracts each cloud once, test/extractor.py:47), for both kinds of pair lists; without it 491 / 548.  Every transfer
    goes from the cloud's owner to a rank whose ranges touch it, and the plan's modelled efficiency rises."""
    from roreg_amd import synth
    from roreg_amd.distributed import shard_scenes, exchange_plan, extractions_per_rank
    names = synth.THREEDMATCH_SCENES
    clouds = dict(zip(names, synth.THREEDMATCH_CLOUDS)); npairs = dict(zip(names, synth.THREEDMATCH_PAIRS))
    cost1 = sum(npairs.values()) + 9.0 * sum(clouds.values())
    for locality, floor in ((None, 0.96), (8.0, 0.97)):
        lists = {s: synth.scene_pair_list(clouds[s], npairs[s], 900 + i, locality=locality) for i, s in enumerate(names)}
        for world in (2, 4, 8):
            plan = shard_scenes(npairs, world, clouds, pair_lists=lists, exchange=True)
            owner, transfers = exchange_plan(plan, lists)
            assert sum(extractions_per_rank(plan, lists)) == 433
            touched = [{(s, int(i)) for s, a, b in r for pr in lists[s][a:b] for i in pr} for r in plan]
            for s, i, src, dst in transfers:
                assert src != dst and owner[(s, i)] == src and (s, i) in touched[src] and (s, i) in touched[dst]
            need = {(c, r) for r in range(world) for c in touched[r] if owner.get(c, r) != r}
            assert need == {((s, i), dst) for s, i, _, dst in transfers}         # exactly what is missing arrives, once
            ex = extractions_per_rank(plan, lists)
            loads = [sum(b - a for _, a, b in r) + 9.0 * ex[q] + 1.0 * sum(1 for t in transfers if t[3] == q) for q, r in enumerate(plan)]
            if world == 8:
                assert cost1 / (world * max(loads)) >= floor, (locality, cost1 / (world * max(loads)))
        old = shard_scenes(npairs, 8, clouds, pair_lists=lists)
        assert sum(extractions_per_rank(old, lists, exchange=False)) > 460


def test_rr_cal_benchmark_matches_reference(tmp_path):
    """Redwood-protocol registration recall (utils/RR_cal.py) on the reference's own inputs/outputs."""
    from types import SimpleNamespace as NS
    from roreg_amd.utils import RR_cal
    z = np.load(os.path.join(GOLDEN, 'rr_cal.npz'))
    scene_dir = tmp_path / 'origin' / 'synth' / 'scene0' / 'PointCloud'
    scene_dir.mkdir(parents=True)
    (scene_dir / 'gt.log').write_bytes(z['gt_log'].tobytes()); (scene_dir / 'gt.info').write_bytes(z['gt_info'].tobytes())
    cfg = NS(output_cache_fn=str(tmp_path / 'cache'), tau_3=0.2)
    pre_dir = tmp_path / 'cache' / 'synth' / 'scene0' / 'match_128' / 'yohoo' / '1000iters'
    pre_dir.mkdir(parents=True)
    (pre_dir / 'pre.log').write_bytes(z['pre_log'].tobytes())
    datasets = {'wholesetname': 'synth', 'scene0': NS(name='synth/scene0', gt_dir=str(scene_dir / 'gt.log'))}
    rr, flags, errors = RR_cal.benchmark(cfg, datasets, 128, 1000, yoho_sign='yohoo')
    assert abs(rr - float(z['rr'])) < 1e-12
    assert np.array_equal(np.array(flags['synth/scene0']), z['flags'])
    assert np.abs(np.array(errors['synth/scene0']) - z['errors']).max() < 1e-9
    got = (tmp_path / 'cache' / 'synth' / 'Eval_results' / 'yohoo_RR' / '1000iters' / 'result.txt').read_text()
    assert got == z['result_txt'].tobytes().decode()


def test_even_groups_of_the_stacked_matcher():
    """engine.even_groups: consecutive, covering, capped (a lone item above the cap is its own group), and even -- the cut that replaced
    32 + 32 + 32 + 4 by 4 x 25."""
    from roreg_amd.engine import even_groups
    assert even_groups([], 10) == []
    assert even_groups([2500] * 100, 80000) == [(0, 25), (25, 50), (50, 75), (75, 100)]
    assert even_groups([2500] * 100, 260000) == [(0, 100)]
    assert even_groups([2500] * 33, 80000) == [(0, 17), (17, 33)]
    assert even_groups([2500, 100, 90000, 2500, 2500], 80000) == [(0, 2), (2, 3), (3, 5)]
    rng = np.random.default_rng(3)
    for _ in range(200):
        sizes = rng.integers(1, 3000, size=int(rng.integers(1, 60))).tolist()
        cap = int(rng.integers(500, 20000))
        g = even_groups(sizes, cap)
        assert g[0][0] == 0 and g[-1][1] == len(sizes) and all(a[1] == b[0] for a, b in zip(g, g[1:])) and all(i < j for i, j in g)
        assert all(sum(sizes[i:j]) <= cap or j - i == 1 for i, j in g)


def test_yohoc_draws_of_many_pairs_in_one_host_call_replay_the_per_pair_streams():
    """roreg_yohoc_draw_many (round 6, host code): the rotation-bin statistic, the sampling loop and the give-up answer of
    `yohoc_draws(anchors, max_iter, np.random.RandomState(seed))` for many pairs in one threaded C call -- the same words of the same MT19937
    streams: rows identical, `rng.rand(4, 4)` of the pairs the reference gives up on identical (test/estimator.py:119-137, 214-230)."""
    from roreg_amd import hip
    from roreg_amd.test.estimator import yohoc_draws
    rng = np.random.default_rng(3)
    anchors, seeds = [], []
    for q in range(40):
        n = int(rng.integers(1, 3000))
        kind = q % 5
        if kind == 0:
            a = rng.integers(0, 60, n)                                            # uniform bins
        elif kind == 1:
            a = np.where(rng.random(n) < 0.7, 17, rng.integers(0, 60, n))         # one dominant rotation
        elif kind == 2:
            a = rng.permutation(60)[:min(n, 60)]                                  # every bin at most one member: the reference gives up
        elif kind == 3:
            a = np.concatenate([np.full(2, 5), rng.permutation(60)[:10]])         # one bin of exactly two (+ maybe one more in bin 5)
        else:
            a = np.full(n, 59)                                                    # everything in the last bin
        anchors.append(np.asarray(a, np.int64)); seeds.append(int(rng.integers(0, 2 ** 32)))
    for max_iter in (1000, 7):
        got, giveup = hip.yohoc_draw_many(seeds, anchors, max_iter, n_threads=3)
        n_gave_up = 0
        for q, (a, sd) in enumerate(zip(anchors, seeds)):
            r = np.random.RandomState(sd)
            want = yohoc_draws(a, max_iter, rng=r)
            if want is None:
                n_gave_up += 1
                assert got[q] is None and np.array_equal(giveup[q], r.rand(4, 4)), q
            else:
                assert got[q] is not None and np.array_equal(got[q], want), q
        assert n_gave_up >= 8


def test_write_npy_files_is_np_save_byte_for_byte(tmp_path):
    """roreg_write_files behind hip.write_npy_files (round 6: the StageFileWriter's per-pair files without np.save's interpreter time): the same
    header bytes numpy writes, the same data -- every dtype and shape of the stage-file contract, empty arrays included."""
    import filecmp
    from roreg_amd import hip
    rng = np.random.default_rng(0)
    arrs = [rng.integers(0, 5000, (137, 2)).astype(np.int64), np.ones(137), rng.random((0, 3, 4)), rng.random((449, 3, 4)), rng.random(5).astype(np.float32),
            np.zeros((0, 2), np.int64), rng.random((50, 32, 60)).astype(np.float32), np.arange(7, dtype=np.int64), np.float32(rng.random(60))[::2]]
    pa = [str(tmp_path / f'a{i}.npy') for i in range(len(arrs))]; pb = [str(tmp_path / f'b{i}.npy') for i in range(len(arrs))]
    for p, a in zip(pa, arrs):
        np.save(p, a)
    hip.write_npy_files(pb, arrs, n_threads=3)
    assert all(filecmp.cmp(x, y, shallow=False) for x, y in zip(pa, pb))
    with pytest.raises(hip.HipError):
        hip.write_npy_files([str(tmp_path / 'no_such_dir' / 'x.npy')], [np.zeros(3)])


def test_result_archives_from_a_template_load_like_np_savez(tmp_path):
    """The evaluator's engine route writes a scene's result archives (test/estimator.py:436-441: np.savez(trans=, [center=,] recalltime=)) from a
    template of np.savez's own bytes with the data and CRCs replaced (_NpzTemplate) through hip.write_files: np.load gives the same members in the
    same order, dtypes and shapes; the archive passes zipfile's CRC check; placeholder files created ahead (StageFileWriter.precreate) are
    overwritten, never kept."""
    import zipfile
    from roreg_amd import hip
    from roreg_amd.test.evaluator import _NpzTemplate
    rng = np.random.default_rng(1)
    for with_center in (False, True):
        make = lambda: {'trans': rng.standard_normal((4, 4)), **({'center': np.ones([6, 3])} if with_center else {}), 'recalltime': int(rng.integers(0, 60000))}
        tpl = _NpzTemplate(make())
        sets = [make() for _ in range(9)]
        assert all(_NpzTemplate.signature(a) == tpl.key for a in sets)
        paths = [str(tmp_path / f'{int(with_center)}-{i}.npz') for i in range(len(sets))]
        for p in paths[:4]:
            open(p, 'wb').write(b'stale bytes of an earlier, longer file' * 100)
        hip.write_files(paths, [tpl.fill(a) for a in sets], n_threads=3)
        for p, a in zip(paths, sets):
            ref = str(tmp_path / 'ref.npz')
            np.savez(ref, **a)
            x, y = np.load(p), np.load(ref)
            assert x.files == y.files
            assert all(np.array_equal(x[k], y[k]) and x[k].dtype == y[k].dtype and x[k].shape == y[k].shape for k in y.files)
            assert zipfile.ZipFile(p).testzip() is None
            assert os.path.getsize(p) == os.path.getsize(ref)
    assert _NpzTemplate.signature({'trans': np.zeros((4, 4)), 'recalltime': np.int32(3)}) != _NpzTemplate.signature({'trans': np.zeros((4, 4)), 'recalltime': 3})


def test_pinned_pool_size_classes():
    """hip.PinnedPool (the buffers of the file loader, the stage-file writer and the staging ring): a class is never smaller than the request,
    wastes at most 12.5 %, and nearby sizes -- a scene's match lists vary by a few percent from scene to scene -- share one."""
    from roreg_amd.hip import PinnedPool
    for n in [1, 4095, 4096, 4097, 65537, 120000, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 38400000, 36 * 10 ** 6, (1 << 30) + 5]:
        c = PinnedPool.size_class(n)
        assert c >= n and (n <= 4096 or c <= n * 1.125 + 1)
        assert PinnedPool.size_class(c) == c
    assert PinnedPool.size_class(35_900_000) == PinnedPool.size_class(36_400_000)
    assert len({PinnedPool.size_class(n) for n in range(1 << 20, 1 << 21, 4099)}) <= 9

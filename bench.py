#!/usr/bin/env python3
"""bench.py -- pair-registrations/sec of the RoReg hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (GF extractor on every cloud, mutual matcher, Des2R + ET local
transforms, one-shot RANSAC + 2 refinements on every pair) over one resident scene chunk of 5000-keypoint clouds
(BASELINE.json configs[1]: one 3DMatch-'kitchen'-like scene, 60-rotation group features, random-init GF/ET
weights of the reference's architecture; data synthetic).  The chunk has the 3DMatch clouds:pairs ratio
(16 clouds, 60 pairs = 0.267 = 433/1623), so `value` already includes the amortised per-cloud work.
With N ranks every rank registers its own chunk (pairs shard with no data-path collective; weak scaling)
and the per-pair result table is all-gathered once per step over RCCL.

The JSON line also carries the roofline of the dominant kernel (the exact-f32 MFMA GEMMs of the group convolution in
the irrep domain, timed with HIP events inside the timed region) and a CPU baseline (the numpy oracle, rank 0,
bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
SUSTAINED_FP16_MFMA_TFLOPS = 1650.0  # measured: tools/mfma_peak.py, random operands, 2-8 waves/SIMD (1840 with one smooth operand pair)
N_KPTS = 5000
N_CLOUDS = 16
N_PAIRS = 60          # 16 clouds : 60 pairs = 0.267 = 3DMatch's 433 clouds : 1623 pairs
OVERLAP = 0.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--kpts', type=int, default=N_KPTS)
    ap.add_argument('--clouds', type=int, default=N_CLOUDS)
    ap.add_argument('--pairs', type=int, default=N_PAIRS)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="torch.distributed backend: 'nccl' (= RCCL, one GPU per rank); "
                    "'gloo' with ROREG_BENCH_SHARED_GPU=1 runs all ranks on GPU 0 -- a control-flow check of the multi-rank path on a 1-GPU box")
    ap.add_argument('--no-secondary', action='store_true', help='skip the all-local-transforms figure (profiling runs)')
    ap.add_argument('--gemm', choices=['f16x2', 'bf16x3', 'f32'], default='f16x2', help="matrix-core mode of the group-conv GEMMs: fp16 x 2 operands with power-of-two block scaling (default), bf16 x 3, or f32-input MFMA; all accumulate in f32")
    return ap.parse_args()


def cpu_baseline(cfg_nets, scene, n_pairs, n_clouds):
    """Oracle (numpy port of the reference's algorithm) timed on a bounded sample, scaled to the step's workload."""
    from oracle import ref_numpy as O
    from roreg_amd.group import tables
    T = tables()
    gf_sd, et_sd = cfg_nets
    threads = torch.get_num_threads()
    t = {}
    nb = 1024
    x = scene.feats[0][:nb]
    t0 = time.perf_counter(); eq = O.gf_forward(x, gf_sd, T.Nei)['eqv']; t['gf_per_kpt'] = (time.perf_counter() - t0) / nb
    n = 3500
    e0 = scene.feats[0][:n]; e1 = scene.feats[1][:n]
    s = np.arange(n)
    t0 = time.perf_counter(); m = O.mutual_match(e0, e1, s, s); dt = time.perf_counter() - t0
    t['mutual_per_pair'] = dt * (scene.feats[0].shape[0] / n) ** 2          # O(N^2)
    nm = 1024
    d1 = scene.feats[1][:nm]; d0 = scene.feats[0][:nm]
    t0 = time.perf_counter(); dr = O.des2r(d1, d0, T.P); t['des2r_per_corr'] = (time.perf_counter() - t0) / nm
    nb = 768
    batch = {'before_eqv0': scene.feats[1][:nb], 'before_eqv1': scene.feats[0][:nb], 'after_eqv0': scene.feats[1][:nb],
             'after_eqv1': scene.feats[0][:nb], 'pre_idx': dr[:nb]}
    t0 = time.perf_counter(); q = O.et_forward(batch, et_sd, T.Nei, T.P); t['et_per_corr'] = (time.perf_counter() - t0) / nb
    M, H = 3000, 200
    rng = np.random.default_rng(0)
    k0 = rng.uniform(0, 3, (M, 3)); k1 = rng.uniform(0, 3, (M, 3)); Tr = rng.standard_normal((H, 3, 4))
    t0 = time.perf_counter()
    for h in range(H):
        O.overlap_cal(k0, k1, Tr[h], np.ones(M), 0.1)
    t['ransac_per_hyp'] = (time.perf_counter() - t0) / H
    return t, threads


DTYPE_OF = {'f16x2': 'f32 (f32-accurate, NOT reduced precision: every f32 operand enters the fp16 matrix cores as hi+lo fp16 under a power-of-two block '
                     'scale = 22 significant bits, 3 cross products, f32 accumulate; measured error <= the f32-input MFMA kernel\'s, 4.0e-7 vs 5.1e-7 '
                     'against the reference; every MFMA kernel of the path); f64 estimator',
            'bf16x3': 'f32 (f32-accurate: every f32 operand as 3 bf16 pieces = 24 bits, 6 cross products, f32 accumulate); f64 estimator',
            'f32': 'f32 (f32-input MFMA, f32 accumulate); f64 estimator'}
MFMAS_PER_PRODUCT = {'f16x2': 3, 'bf16x3': 6}
KERNEL_OF = {'f16x2': 'irrep_gemm_split_kernel<32,2> (GF 256->512 / 512->256 in the irrep domain, fp16 x 2 block-scaled operands: 3 fp16 MFMAs per product)',
             'bf16x3': 'irrep_gemm_split_kernel<32,3> (GF 256->512 / 512->256 in the irrep domain, 3 x bf16 split operands: 6 bf16 MFMAs per product)',
             'f32': 'irrep_gemm_kernel<32> (GF 256->512 / 512->256 in the irrep domain, f32-input MFMA)'}


def roofline_obj(mode, gemm_tflops, ms, n_launch, traffic):
    """MFMA roofline of the dominant kernel, in EXECUTED matrix-core flops: the kernel performs, per irrep, the GEMM
    [d*O x d*C] . [d*C x d*B], i.e. 2*O*C*B*244 flop per launch = `gemm_tflops` when divided by its time.
    'f32': v_mfma_f32_32x32x2_f32, priced against the f32-input MFMA peak.  'bf16x3' / 'f16x2': every product is six bf16 / three fp16
    MFMAs, so 6x / 3x gemm_tflops are executed and priced against the dense bf16 = fp16 peak; the f32-equivalent rate is reported too.
    (In the reference's own 13-stencil form the same layer is 780/244 = 3.2x more flops: SURVEY 8d's per-keypoint figure.)"""
    base = {'bound': 'mfma', 'unit': 'TFLOP/s', 'avg_launch_ms': ms / max(n_launch, 1), 'launches': n_launch, 'traffic': traffic,
            'f32_equivalent_gemm_tflops': gemm_tflops, 'reference_stencil_form_equivalent_tflops': gemm_tflops * 780.0 / 244.0}
    if mode == 'f32':
        base.update({'kernel': KERNEL_OF[mode], 'achieved': gemm_tflops, 'peak': PEAK_F32_MFMA_TFLOPS, 'frac': gemm_tflops / PEAK_F32_MFMA_TFLOPS})
    else:
        k = MFMAS_PER_PRODUCT[mode]
        base.update({'kernel': KERNEL_OF[mode], 'achieved': k * gemm_tflops, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': k * gemm_tflops / PEAK_BF16_MFMA_TFLOPS,
                     # context, not the priced peak: what back-to-back 32x32x16 fp16 MFMAs from registers sustain on this part with
                     # pseudo-random operands (tools/mfma_peak.py) -- the power-limited ceiling of any 16-bit kernel
                     'sustained_mfma_ceiling_measured_tflops': SUSTAINED_FP16_MFMA_TFLOPS,
                     'frac_of_sustained_ceiling': k * gemm_tflops / SUSTAINED_FP16_MFMA_TFLOPS})
    return base


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(0 if os.environ.get('ROREG_BENCH_SHARED_GPU') else local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    coll_dev = 'cuda' if args.backend == 'nccl' else 'cpu'                 # device of the (tiny) collective payloads

    from roreg_amd import hip, synth
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config

    cfg = default_config(keynum=args.kpts, max_iter=1000, ET='yohoo')
    gf = name2network['GF_test'](cfg); gf_sd = synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); et_sd = synth.seeded_state_dict(et, 202)
    eng = RegistrationEngine(cfg, gf, et)
    eng.set_gemm_mode(args.gemm)

    scene = synth.make_scene(1000 + rank, n_clouds=args.clouds, n_kpts=args.kpts, overlap=OVERLAP, coord_noise=0.005)
    # a fixed pseudo-random subset of the cloud pairs that touches every cloud (like a scene's gt.log lists the overlapping pairs)
    order = np.random.default_rng(4242).permutation(len(scene.pair_ids))
    pair_ids = [scene.pair_ids[i] for i in sorted(order[:min(args.pairs, len(order))])]
    feats = [torch.from_numpy(f).cuda() for f in scene.feats]            # inputs resident in HBM before timing
    keys = [torch.from_numpy(k).cuda() for k in scene._kps]
    n_pairs = len(pair_ids)

    def step(all_lt=False):
        np.random.seed(7)
        res = eng.run_scene(feats, keys, pair_ids, all_local_transforms=all_lt)
        table = torch.tensor([[float(r.id0), float(r.id1), r.n_match, r.recalltime] + r.trans.reshape(-1).tolist() for r in res],
                             dtype=torch.float64, device=coll_dev)
        if dist is not None:
            out = [torch.empty_like(table) for _ in range(world)]
            dist.all_gather(out, table)                                     # the single result-table collective (RCCL/xGMI)
        return res

    for _ in range(args.warmup):
        step()

    hip.PROFILE = []                                                         # per-launch HIP events of the group conv
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = hip.PROFILE; hip.PROFILE = None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    def timed(n, **kw):
        """n timed steps (after one untimed one) with the same barrier + synchronise bracket; MAX over ranks."""
        if n == 0:
            return 0.0, res
        step(**kw)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            r = step(**kw)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        d = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([d], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d, r

    # ---- secondary figure 1: the same steps with the local transform of EVERY correspondence evaluated, as the reference's
    # file-coupled estimator does (the default evaluates only the <= max_iter hypotheses one-shot RANSAC draws; same results) ----
    n_all = 0 if args.no_secondary else max(1, args.steps // 2)
    dt_all, res_all = timed(n_all, all_lt=True)
    same = all(np.array_equal(a.trans, b.trans) and a.recalltime == b.recalltime for a, b in zip(res, res_all))

    # ---- secondary figure 2: the other GEMM mode (3 x bf16 split operands, f32 accumulate: f32-accurate, not bit-equal) ----
    other = 'f32' if args.gemm != 'f32' else 'f16x2'
    eng.set_gemm_mode(other)
    hip.PROFILE = []
    dt_other, res_other = timed(n_all)
    prof_other = hip.PROFILE; hip.PROFILE = None
    eng.set_gemm_mode(args.gemm)
    max_dT = max([float(np.abs(a.trans - b.trans).max()) for a, b in zip(res, res_other) if np.isfinite(a.trans).all() and np.isfinite(b.trans).all()] or [0.0])

    # ---- secondary figure 3: the split SURVEY 8(d) asks for -- per-cloud stages (extractor) vs per-pair stages (matcher, local
    # transforms, estimator) -- from one extra step with a synchronisation after every phase (diagnostic, outside the timed region) ----
    phases = None
    if not args.no_secondary:
        eng.phase_ms = {}
        step()
        phases = {k: round(v, 2) for k, v in eng.phase_ms.items()}
        eng.phase_ms = None

    # ---- roofline of the dominant kernel: the irrep-domain GEMMs of the two big GF layers (256->512, 512->256) ----
    # algorithmic work per launch = sum over the five irreps of 2*(d*O)*(d*C)*(d*B) = 2*O*C*B*244 flop
    def gemm_roofline(events, tagname):
        flops = 0.0; ms = 0.0; n_launch = 0
        for (tag, e0, e1) in events:
            if tag[0] == tagname and tag[2] * tag[3] == 256 * 512:
                _, B_, C_, O_ = tag
                flops += 2.0 * O_ * C_ * B_ * 244
                ms += e0.elapsed_time(e1); n_launch += 1
        return (flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), ms, n_launch

    tag_of = {'f32': 'irrep_gemm', 'bf16x3': 'irrep_gemm_split', 'f16x2': 'irrep_gemm_f16x2'}
    achieved, ms, n_launch = gemm_roofline(prof, tag_of[args.gemm])
    achieved_other, ms_other, n_other = gemm_roofline(prof_other or [], tag_of[other])
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'r01_irrep_gemm_pmc.json')
    if os.path.exists(pmc):                      # HBM bytes per launch from the rocprofv3 --pmc passes of this same command
        traffic = json.load(open(pmc)).get('hbm_bytes_per_launch', {}).get(args.gemm) if isinstance(json.load(open(pmc)).get('hbm_bytes_per_launch'), dict) else None

    # ---- accuracy on the synthetic chunk (outside the timed region; the evaluator's own metric code) ----
    from roreg_amd.utils.r_eval import compute_R_diff
    rr = []
    for r in res:
        gt = scene.get_transform(r.id0, r.id1)
        if np.isfinite(r.trans).all():
            rd = compute_R_diff(r.trans[:3, :3], gt[:3, :3]); td = float(np.sqrt(np.sum(np.square(r.trans[:3, 3] - gt[:3, 3]))))
            rr.append(1 if (rd < 15 and td < 0.3) else 0)
        else:
            rr.append(0)

    if rank == 0:
        value = world * n_pairs * args.steps / dt
        out = {
            'metric': 'pair-registrations/sec', 'value': value, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE_OF[args.gemm], 'data': 'synthetic',
            'config': {'workload': f"3DMatch-kitchen-like scene chunk per GPU: {args.clouds} clouds x {args.kpts} kpts, {n_pairs} pairs "
                                   f"(mutual matcher + yohoo estimator, max_iter=1000, 60-rot group feats)",
                       'pairs_per_step_per_gpu': n_pairs, 'clouds_per_step_per_gpu': args.clouds, 'parallelism': f'pairs-sharded x{world}',
                       'mean_matches': float(np.mean([r.n_match for r in res])), 'registration_recall_synthetic': float(np.mean(rr)),
                       'local_transforms': 'only the <=1000 hypotheses one-shot RANSAC draws per pair (results identical to evaluating all M)',
                       'value_all_local_transforms': (world * n_pairs * n_all / dt_all) if n_all else None, 'results_identical_to_all_local_transforms': bool(same),
                       'phase_ms_one_synchronised_step': phases,
                       'pair_stages_only_pairs_per_s_per_gpu': (n_pairs / (sum(v for k, v in phases.items() if k != 'extract') * 1e-3)) if phases else None,
                       'per_cloud_stage_clouds_per_s_per_gpu': (args.clouds / (phases['extract'] * 1e-3)) if phases else None},
            'roofline': roofline_obj(args.gemm, achieved, ms, n_launch, traffic),
        }
        if n_all:
            out['config']['other_gemm_mode'] = {
                'mode': other, 'dtype': DTYPE_OF[other], 'value': world * n_pairs * n_all / dt_other, 'ms_per_step': 1e3 * dt_other / n_all,
                'max_abs_diff_of_transforms_vs_default': max_dT,
                'roofline': roofline_obj(other, achieved_other, ms_other, n_other, None)}
        if not args.no_cpu_baseline and world == 1:
            gf_np = {k: v.numpy() for k, v in gf_sd.items()}; et_np = {k: v.numpy() for k, v in et_sd.items()}
            t, threads = cpu_baseline((gf_np, et_np), scene, n_pairs, args.clouds)
            M = float(np.mean([r.n_match for r in res])); H = min(M, 1000)
            per_cloud = t['gf_per_kpt'] * args.kpts
            per_pair = t['mutual_per_pair'] + M * (t['des2r_per_corr'] + t['et_per_corr']) + H * t['ransac_per_hyp'] * (M / 3000.0)
            sec = args.clouds * per_cloud + n_pairs * per_pair
            out['cpu_baseline'] = {'value': n_pairs / sec, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
                                   'sample': 'oracle/ref_numpy.py on this host: GF on 1024 kpts, mutual on 3500x3500 (scaled N^2), Des2R on 1024, '
                                             'ET on 768 correspondences, RANSAC scoring on 200 hypotheses x 3000; scaled to the step workload',
                                   'components_s': {'gf_per_cloud': per_cloud, 'per_pair': per_pair}}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

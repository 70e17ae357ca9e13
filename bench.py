#!/usr/bin/env python3
"""bench.py -- pair-registrations/sec of the RoReg hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload 3dmatch-full | kitchen | chunk] [--gemm f16x2 | bf16x3 | f32]
                    [--dtype fp32 | bf16] [--pair-lists banded | uniform] [--no-exchange]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: it starts N ranks of itself through
torch.distributed.run (127.0.0.1 rendezvous) before anything touches a GPU, relays rank 0's JSON line and exits with the job's
code.  Every rank checks that the world size it sees equals --gpus and fails otherwise.

A "step" is one pass of the whole hot path (GF extractor on every cloud, mutual matcher, Des2R + ET local transforms, one-shot
RANSAC + 2 refinements on every pair) over the workload, inputs resident in HBM before the timed region:

  3dmatch-full (default, every N): the full 3DMatch test shape (BASELINE.json configs[2]) -- 8 synthetic scenes with the benchmark's
           station counts [60,60,60,55,57,37,66,38] (dataops/dataset.py:152) = 433 clouds x 5000 keypoints and 1623 pairs (kitchen: 60 clouds,
           449 pairs; pair lists: the chain (i, i+1) plus pairs drawn with probability ~ exp(-|i-j|/8), --pair-lists uniform for uniformly
           drawn ones), random-init GF/ET weights of the reference's architecture.  With N ranks the pairs are sharded by
           roreg_amd.distributed.shard_scenes (whole scenes first, the scenes that must be cut into pair ranges); every cloud is extracted
           by ONE rank and a cut scene's extractor outputs travel point to point (RCCL send/recv over xGMI) at the start of the step;
           ONE all_gather of the result table per step: STRONG scaling, the same command for every N.  At N = 1 the whole benchmark runs
           on one GPU (it fits: 16.6 GB of inputs); the kitchen scene alone (configs[1]) is timed in passes of its own and reported as
           `config.kitchen_scene`.
  kitchen: only the kitchen-like scene (60 clouds, 449 pairs), pairs sharded across the ranks.
  chunk  : round 1's 16-cloud / 60-pair scene chunk per rank (weak scaling), kept for comparison.

The JSON line carries: the roofline of the dominant kernel (the fp16x2 MFMA GEMMs of the group convolution in the irrep domain, HIP events
on the launch stream inside the timed region); measured rooflines of the kernels north_star names (descriptor distance matrix, RANSAC
scoring, Des2R); `value_all_local_transforms` (the reference's per-pair work: the local transform of EVERY correspondence) on the same full
workload; FMR / IR / RR over all ranks from the evaluator's metric code; a secondary leg on BASELINE configs[3]/[4]'s path (detector +
rotation-coherence matcher at keynum 2500, float32 and bfloat16 descriptor storage) with the rooflines of its Sinkhorn and top-k kernels;
and two CPU baselines on the host cores (the reference's own torch CPU operators on all threads = oracle/ref_torch.py, and the numpy
oracle), rank 0, bounded samples.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense f32-input MFMA peak (= the f32 vector FMA peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (no sparsity)
PEAK_F64_VALU_TFLOPS = 78.6         # MI355X_MICROARCH.md: vector fp64
PEAK_HBM_GBS = 8000.0
N_KPTS = 5000
OVERLAP = 0.6
STUB = os.environ.get('ROREG_BENCH_ENGINE')          # tests only: 'module:factory' of a host-side stand-in engine (tests/_bench_stub.py)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='3dmatch-full', choices=['3dmatch-full', 'kitchen', 'chunk'])
    ap.add_argument('--kpts', type=int, default=N_KPTS)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="torch.distributed backend: 'nccl' (= RCCL, one GPU per rank); "
                    "'gloo' with ROREG_BENCH_SHARED_GPU=1 runs all ranks on GPU 0 -- a control-flow check of the multi-rank path on a 1-GPU box")
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary figures (profiling runs)')
    ap.add_argument('--gemm', choices=['f16x2', 'bf16x3', 'f32'], default='f16x2', help="matrix-core mode of the group-conv GEMMs: fp16 x 2 operands with "
                    "per-keypoint power-of-two block scaling (default), bf16 x 3, or f32-input MFMA; all accumulate in f32")
    ap.add_argument('--pair-lists', choices=['banded', 'uniform'], default='banded', help='synthetic pair lists of a scene: beyond the chain (i, i+1), pair (i, j) is '
                    'drawn with probability ~ exp(-|i-j|/8) (default: overlapping fragments of a scan sequence are mostly temporally close) or uniformly over '
                    'all cloud pairs (the worst case for cutting a scene across ranks); same pair and cloud counts, same single-GPU work')
    ap.add_argument('--dtype', choices=['fp32', 'bf16'], default='fp32', help='descriptor storage (BASELINE config 5: bf16 = group features stored and '
                    'streamed as bfloat16, float32 accumulation)')
    ap.add_argument('--no-exchange', action='store_true', help='multi-rank: every rank extracts every cloud its pair ranges touch (round 2) instead of '
                    'extracting each cloud once and shipping the extractor output of a cut scene over xGMI')
    ap.add_argument('--all-steps', type=int, default=5, help='timed steps of the all-local-transforms figure (the reference\'s per-pair work)')
    ap.add_argument('--master-port', type=int, default=0, help='launcher only: rendezvous port (0 = pick a free one)')
    ap.add_argument('--force-collectives', action='store_true', help='N = 1: initialise the process group (nccl = RCCL, one rank) anyway, run the result '
                    'table\'s all_gather every step and ship the extractor outputs of 4 clouds from the rank to itself through the grouped send/recv: a '
                    'single GPU then executes the whole multi-GPU code path (ROREG_FORCE_COLLECTIVES=1 does the same)')
    ap.add_argument('--pipeline', choices=['mutual', 'rd_rm'], default='mutual', help="which pipeline the headline region times: 'mutual' (mutual matcher + yohoo: "
                    "the default) or 'rd_rm' (`Test.py --RD --RM --ET yohoo --keynum 5000`: detector + NMS + rotation-coherence matcher + yohoo; the default run "
                    "reports it as value_rd_rm_k5000 -- this switch makes it the timed region itself, for profiling)")
    ap.add_argument('--rd-rm-steps', type=int, default=2, help='timed steps of RoReg\'s own pipeline (--RD --RM --ET yohoo at --keynum = --kpts) on the full workload (0 = skip)')
    ap.add_argument('--no-dropin', action='store_true', help='skip the drop-in leg (one scene through yoho_evaluator.process_scene on disk: ~7 GB of temporary files)')
    ap.add_argument('--bf16x3-steps', type=int, default=2, help='timed steps of the strictly 24-bit matrix-core mode on the full workload (0 = skip)')
    return ap.parse_args(argv)


# ---- launcher ---------------------------------------------------------------------------------------------------------------------------
def launch(args):
    """--gpus N > 1 without a torch.distributed environment: start N ranks of this script (one process per GPU) through
    torch.distributed.run.  Runs before any GPU call of this process -- a process that has initialised the GPU must never be replaced or
    fork workers -- and the children are ordinary subprocesses; their exit code is ours."""
    import socket
    shared = bool(os.environ.get('ROREG_BENCH_SHARED_GPU')) or bool(STUB)
    n_dev = torch.cuda.device_count()                       # (counting devices does not initialise the GPU)
    if not shared and n_dev < args.gpus:
        print(f'bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible (ROREG_BENCH_SHARED_GPU=1 --backend gloo runs every rank on GPU 0)', file=sys.stderr)
        return 2
    port = args.master_port
    if not port:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '8'))
    return subprocess.call(cmd, env=env)


# ---- workload ---------------------------------------------------------------------------------------------------------------------------
def build_workload(args, rank, world, device='cuda', exchange=True):
    """-> (scenes {name: (feats, keys, poses, pair_ids)} for the scenes this rank touches, plan [(scene, a, b)] of this rank, totals)."""
    from roreg_amd import synth
    from roreg_amd.distributed import shard_scenes, exchange_plan, extractions_per_rank
    if args.workload == 'chunk':
        feats, keys, poses = synth.make_scene_device(1000 + rank, 16, args.kpts, OVERLAP, device=device)
        pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(16, 60, 4242)]
        return {'chunk': (feats, keys, poses, pairs)}, [('chunk', 0, len(pairs))], {'pairs': 60 * world, 'clouds': 16 * world, 'scaling': 'weak', 'transfers': [],
                                                                                   'rows_per_rank': [60] * world}
    names = synth.THREEDMATCH_SCENES if args.workload == '3dmatch-full' else synth.THREEDMATCH_SCENES[:1]
    clouds = dict(zip(synth.THREEDMATCH_SCENES, synth.THREEDMATCH_CLOUDS)); npairs = dict(zip(synth.THREEDMATCH_SCENES, synth.THREEDMATCH_PAIRS))
    loc = 8.0 if getattr(args, 'pair_lists', 'banded') == 'banded' else None
    lists = {s: synth.scene_pair_list(clouds[s], npairs[s], 900 + synth.THREEDMATCH_SCENES.index(s), locality=loc) for s in names}
    exchange = bool(exchange and world > 1)
    plan = shard_scenes({s: npairs[s] for s in names}, world, {s: clouds[s] for s in names}, pair_lists=lists, exchange=exchange)
    transfers = exchange_plan(plan, lists)[1] if exchange else []
    scenes = {}
    for s in sorted({p[0] for p in plan[rank]} | {t[0] for t in transfers if rank in (t[2], t[3])}):
        i = synth.THREEDMATCH_SCENES.index(s)
        feats, keys, poses = synth.make_scene_device(500 + i, clouds[s], args.kpts, OVERLAP, device=device)
        scenes[s] = (feats, keys, poses, [(str(a), str(b)) for a, b in lists[s]])
    return scenes, plan[rank], {'pairs': sum(npairs[s] for s in names), 'clouds': sum(clouds[s] for s in names), 'scaling': 'strong',
                                'plan': [[list(p) for p in r] for r in plan], 'transfers': transfers,
                                'rows_per_rank': [sum(b - a for _, a, b in r) for r in plan],
                                'extractions_per_rank': extractions_per_rank(plan, lists, exchange=exchange)}


# ---- CPU baselines ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline_numpy(cfg_nets, feats0, feats1):
    """Oracle (numpy port of the reference's algorithm) timed on a bounded sample (per-unit costs)."""
    from oracle import ref_numpy as O
    from roreg_amd.group import tables
    T = tables()
    gf_sd, et_sd = cfg_nets
    t = {}
    nb = 512
    x = feats0[:nb]
    t0 = time.perf_counter(); O.gf_forward(x, gf_sd, T.Nei); t['gf_per_kpt'] = (time.perf_counter() - t0) / nb
    n = 2500
    e0 = feats0[:n]; e1 = feats1[:n]
    s = np.arange(n)
    t0 = time.perf_counter(); O.mutual_match(e0, e1, s, s); dt = time.perf_counter() - t0
    t['mutual_per_pair'] = dt * (feats0.shape[0] / n) ** 2                  # O(N^2)
    nm = 512
    d1 = feats1[:nm]; d0 = feats0[:nm]
    t0 = time.perf_counter(); dr = O.des2r(d1, d0, T.P); t['des2r_per_corr'] = (time.perf_counter() - t0) / nm
    nb = 384
    batch = {'before_eqv0': feats1[:nb], 'before_eqv1': feats0[:nb], 'after_eqv0': feats1[:nb], 'after_eqv1': feats0[:nb], 'pre_idx': dr[:nb]}
    t0 = time.perf_counter(); O.et_forward(batch, et_sd, T.Nei, T.P); t['et_per_corr'] = (time.perf_counter() - t0) / nb
    M, H = 3000, 200
    rng = np.random.default_rng(0)
    k0 = rng.uniform(0, 3, (M, 3)); k1 = rng.uniform(0, 3, (M, 3)); Tr = rng.standard_normal((H, 3, 4))
    t0 = time.perf_counter()
    for h in range(H):
        O.overlap_cal(k0, k1, Tr[h], np.ones(M), 0.1)
    t['ransac_per_hyp'] = (time.perf_counter() - t0) / H
    return t


def cpu_baseline_torch(cfg_nets, feats0, feats1):
    """The reference's own torch CPU operators (oracle/ref_torch.py: conv2d on the gathered stencil tensor, batch_norm, einsum, the chunked
    nearest-neighbour search) on ALL host threads, bounded samples (SURVEY 8d's CPU baseline) -> per-unit costs."""
    from oracle import ref_torch as OT
    from roreg_amd.group import tables
    T = tables()
    gf_sd, et_sd = cfg_nets
    f0 = torch.from_numpy(feats0); f1 = torch.from_numpy(feats1)
    t = {}
    nb = 1250                                                               # the reference's bs_GF (test/extractor.py:51)
    OT.gf_forward(f0[:64], gf_sd, T.Nei)                                    # (operator warm-up: oneDNN primitive creation)
    t0 = time.perf_counter(); OT.gf_forward(f0[:nb], gf_sd, T.Nei); t['gf_per_kpt'] = (time.perf_counter() - t0) / nb
    n = feats0.shape[0]
    s = np.arange(n)
    t0 = time.perf_counter(); OT.mutual_match(f0, f1, s, s); t['mutual_per_pair'] = time.perf_counter() - t0       # the full 5000 x 5000 search
    nm = 1000
    t0 = time.perf_counter(); dr = OT.des2r(f1[:nm], f0[:nm], T.P); t['des2r_per_corr'] = (time.perf_counter() - t0) / nm
    nb = 1000                                                               # bs_ET (test/estimator.py:338)
    batch = {'before_eqv0': f1[:nb], 'before_eqv1': f0[:nb], 'after_eqv0': f1[:nb], 'after_eqv1': f0[:nb], 'pre_idx': dr[:nb]}
    OT.et_forward({k: v[:64] for k, v in batch.items()}, et_sd, T.Nei, T.P)
    t0 = time.perf_counter(); OT.et_forward(batch, et_sd, T.Nei, T.P); t['et_per_corr'] = (time.perf_counter() - t0) / nb
    M, H = 3000, 200
    rng = np.random.default_rng(0)
    k0 = rng.uniform(0, 3, (M, 3)); k1 = rng.uniform(0, 3, (M, 3)); Tr = rng.standard_normal((H, 3, 4))
    t0 = time.perf_counter()
    for h in range(H):
        OT.overlap_cal(k0, k1, Tr[h], np.ones(M), 0.1)
    t['ransac_per_hyp'] = (time.perf_counter() - t0) / H
    return t


def scale_baseline(t, M, clouds, pairs):
    """per-unit host costs -> (pairs/s with the local transform of EVERY correspondence = the reference's work, pairs/s with only the drawn
    hypotheses' = what the headline `value` does, components)."""
    H = min(M, 1000)
    per_cloud = t['gf_per_kpt'] * N_KPTS
    tail = H * t['ransac_per_hyp'] * (M / 3000.0)
    per_pair_all = t['mutual_per_pair'] + M * (t['des2r_per_corr'] + t['et_per_corr']) + tail
    per_pair_drawn = t['mutual_per_pair'] + H * (t['des2r_per_corr'] + t['et_per_corr']) + tail
    return (pairs / (clouds * per_cloud + pairs * per_pair_all), pairs / (clouds * per_cloud + pairs * per_pair_drawn),
            {'gf_per_cloud': per_cloud, 'per_pair_all_local_transforms': per_pair_all, 'per_pair_drawn_hypotheses_only': per_pair_drawn})


DTYPE_OF = {'f16x2': 'f32 (every f32 operand enters the fp16 matrix cores as hi + lo fp16 under a per-keypoint power-of-two block scale: <= 22 significant bits '
                     'relative to a CONSERVATIVE per-keypoint bound -- measured on the extractor: the bound sits 2^2..2^6 above the keypoint\'s true coefficient '
                     'maximum (config.f16x2_scale_headroom_bits, this run), i.e. 16..20 bits relative to the data plus the lo piece\'s absolute floor of 2^-39 of the bound; 3 cross '
                     'products, f32 accumulate; measured GEMM error 7e-7..9e-7 of the output scale vs 1.5e-6..2e-6 for the f32-input MFMA kernel; '
                     'results independent of batch composition and rank count); f64 estimator',
            'bf16x3': 'f32 (f32-accurate: every f32 operand as 3 bf16 pieces = 24 bits, 6 cross products, f32 accumulate); f64 estimator',
            'f32': 'f32 (f32-input MFMA, f32 accumulate); f64 estimator'}
MFMAS_PER_PRODUCT = {'f16x2': 3, 'bf16x3': 6}
KERNEL_OF = {'f16x2': 'irrep_gemm_xdma16_kernel<1> (8-wave 256 x 256 tile, v_mfma_f32_16x16x32_f16 on K = 32 steps, activations and weights global -> LDS by DMA into rings '
                      'that take the whole LDS, LDS -> fragments by transposing reads, epilogue through LDS with 16-byte stores; GF 256->512 / 512->256 in the '
                      'irrep domain, fp16 x 2 operands pre-split by ft_nonlin under per-keypoint block scales: 3 fp16 MFMAs per product)',
             'f16x2-mfma32': 'irrep_gemm_xdma_kernel<1> (ROREG_GEMM_MFMA16=0: the same operands and LDS images under v_mfma_f32_32x32x16_f16, K = 16 steps; bitwise the register-staged kernel)',
             'bf16x3': 'irrep_gemm_split_kernel<32,3,2,1> (GF 256->512 / 512->256 in the irrep domain, 3 x bf16 split operands: 6 bf16 MFMAs per product)',
             'f16x2-words': 'irrep_gemm_split_kernel<32,2,4,1,1> (ROREG_GEMM_XDMA=0: fragment-pipelined 8-wave loop with the activations staged through registers; GF 256->512 / '
                            '512->256 in the irrep domain, fp16 x 2 operands pre-split by ft_nonlin under per-keypoint block scales: 3 fp16 MFMAs per product)',
             'f32': 'irrep_gemm_kernel<32> (GF 256->512 / 512->256 in the irrep domain, f32-input MFMA)'}
TAG_OF = {'f32': 'irrep_gemm', 'bf16x3': 'irrep_gemm_split', 'f16x2': 'irrep_gemm_f16x2'}


def roofline_obj(mode, gemm_tflops, ms, n_launch, traffic, alg_bytes):
    """MFMA roofline of the dominant kernel, in EXECUTED matrix-core flops: the kernel performs, per irrep, the GEMM
    [d*O x d*C] . [d*C x d*B], i.e. 2*O*C*B*244 flop per launch = `gemm_tflops` when divided by its time.
    'f32': v_mfma_f32_32x32x2_f32, priced against the f32-input MFMA peak.  'bf16x3' / 'f16x2': every product is six bf16 / three fp16
    MFMAs, so 6x / 3x gemm_tflops are executed and priced against the dense bf16 = fp16 peak; the f32-equivalent rate is reported too.
    (In the reference's own 13-stencil form the same layer is 780/244 = 3.2x more flops: SURVEY 8d's per-keypoint figure.)"""
    base = {'bound': 'mfma', 'unit': 'TFLOP/s', 'avg_launch_ms': ms / max(n_launch, 1), 'launches': n_launch,
            'traffic': None if traffic is None else traffic.get('bytes'), 'traffic_source': traffic,
            'mfma_busy_fraction': None if traffic is None else traffic.get('mfma_busy_fraction'),      # SQ_VALU_MFMA_BUSY_CYCLES of the same PMC pass
            'traffic_over_algorithmic': (traffic.get('bytes') / alg_bytes) if (traffic and traffic.get('bytes') and alg_bytes) else None,
            'algorithmic_bytes_per_launch': alg_bytes, 'f32_equivalent_gemm_tflops': gemm_tflops,
            'reference_stencil_form_equivalent_tflops': gemm_tflops * 780.0 / 244.0}
    if mode == 'f32':
        base.update({'kernel': KERNEL_OF[mode], 'achieved': gemm_tflops, 'peak': PEAK_F32_MFMA_TFLOPS, 'frac': gemm_tflops / PEAK_F32_MFMA_TFLOPS})
    else:
        k = MFMAS_PER_PRODUCT[mode]
        from roreg_amd import hip
        base.update({'kernel': KERNEL_OF[('f16x2-words' if not hip.use_planes(512) else 'f16x2' if hip.MFMA16 else 'f16x2-mfma32') if mode == 'f16x2' else mode], 'achieved': k * gemm_tflops, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': k * gemm_tflops / PEAK_BF16_MFMA_TFLOPS})
    return base


def gemm_roofline(events, tagname):
    """(f32-equivalent TFLOP/s, total ms, launches, mean algorithmic bytes) of the big-layer GEMM launches (C * O = 256 * 512)."""
    flops = 0.0; ms = 0.0; n_launch = 0; alg = 0.0
    for (tag, e0, e1) in events:
        if tag[0] == tagname and tag[2] * tag[3] == 256 * 512:
            _, B_, C_, O_ = tag
            flops += 2.0 * O_ * C_ * B_ * 244
            alg += 60.0 * B_ * 4 * (C_ + O_ + (O_ if C_ == 512 else 0))         # X + Out (+ the short-cut operand of 512 -> 256)
            ms += e0.elapsed_time(e1); n_launch += 1
    return (flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), ms, n_launch, (alg / n_launch if n_launch else None)


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the dominant kernel's source file: a PMC pass is only quoted for the source it was collected on."""
    with open(os.path.join(ROOT, 'roreg_amd', 'csrc', 'fourier.hip'), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


PMC_DEFAULT_CONDITIONS = {'kpts': 5000, 'dtype': 'fp32', 'gpus': 1, 'pair_lists': 'banded', 'xdma': True, 'mfma16': True}


def measured_traffic(args, world=1):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/r*_irrep_gemm_pmc.json) -- returned
    only if that pass was collected on THIS kernel source (its `kernel_source_sha16` equals csrc/fourier.hip's hash now) AND under this
    run's conditions (keypoints per cloud, descriptor storage type, rank count, pair lists, the LDS-DMA kernel switch: they decide the
    launch sizes and which kernel runs); otherwise None with the reason, so a number is never printed for a run it was not measured on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_irrep_gemm_pmc.json')))
    if not files:
        return None
    j = json.load(open(files[-1]))
    val = (j.get('hbm_bytes_per_launch') or {}).get(f'{args.workload}:{args.gemm}')
    src = {'file': os.path.relpath(files[-1], ROOT), 'git_commit': j.get('git_commit'), 'kernel_source_sha16': j.get('kernel_source_sha16')}
    if val is None:
        return None
    if j.get('kernel_source_sha16') != kernel_source_hash():
        return {'bytes': None, **src, 'refused': f'the PMC pass was collected on another kernel source (now {kernel_source_hash()})'}
    from roreg_amd import hip
    now = {'kpts': args.kpts, 'dtype': args.dtype, 'gpus': world, 'pair_lists': args.pair_lists, 'xdma': hip.use_planes(512), 'mfma16': bool(hip.MFMA16)}
    then = dict(PMC_DEFAULT_CONDITIONS, **(j.get('conditions') or {}))
    diff = {k: (then.get(k), now[k]) for k in now if then.get(k) != now[k]}
    if diff:
        return {'bytes': None, **src, 'refused': f'the PMC pass was collected under other conditions (then, now): {diff}'}
    det = (j.get('detail') or {}).get(f'{args.workload}:{args.gemm}') or {}
    return {'bytes': val, **src, 'conditions': then, 'mfma_busy_fraction': det.get('mfma_busy_fraction')}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch(args))
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    # stdout carries ONE line, the JSON record: everything else this process or its libraries write to file descriptor 1 (RCCL prints a version
    # banner there when a communicator is created) goes to stderr; the record is written to the saved descriptor at the end
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)', file=sys.stderr)
        sys.exit(3)
    if STUB:
        device = 'cpu'; dev_index = None; args.backend = 'gloo'
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
        dev_index = 0 if os.environ.get('ROREG_BENCH_SHARED_GPU') else local
        torch.cuda.set_device(dev_index)
        device = 'cuda'
    sync = (lambda: None) if STUB else torch.cuda.synchronize
    from roreg_amd import distributed as D
    if args.force_collectives:
        os.environ['ROREG_FORCE_COLLECTIVES'] = '1'
    forced = D.forced()
    dist = None
    if world > 1 or forced:
        dist = D.init_collectives(args.backend, rank, world, dev_index)      # init_process_group + the group's first barrier, under the hang watchdog
    coll_dev = 'cuda' if (args.backend == 'nccl' and not STUB) else 'cpu'    # device of the (tiny) collective payloads
    step_timeout = D.collective_timeout(180.0)                               # a step is ~1.5 s; the first one also builds RCCL's peer connections

    def barrier():
        if dist is not None:
            with D.watchdog(step_timeout, 'barrier'):
                dist.barrier(device_ids=[dev_index]) if args.backend == 'nccl' else dist.barrier()

    from roreg_amd import synth
    from roreg_amd.parses.parses_test import default_config
    cfg = default_config(keynum=args.kpts, max_iter=1000, ET='yohoo')
    eng_rr = rr_weights = None
    if STUB:
        import importlib
        mod, fn = STUB.split(':')
        eng = getattr(importlib.import_module(mod), fn)(cfg)
        hip = None
    else:
        from roreg_amd import hip
        from roreg_amd.engine import RegistrationEngine
        from roreg_amd.network import name2network
        gf = name2network['GF_test'](cfg); gf_sd = synth.seeded_state_dict(gf, 101)
        et = name2network['ET_test'](cfg); et_sd = synth.seeded_state_dict(et, 202)
        eng = RegistrationEngine(cfg, gf, et)
        eng.set_gemm_mode(args.gemm)
        if args.dtype == 'bf16':
            eng.set_descriptor_dtype('bf16')
        # RoReg's own pipeline (README.md:149,175: `Test.py --RD --RM --ET yohoo --keynum 5000`) on the same workload: a second engine over the same
        # extractor / ET networks plus the detector and the rotation-coherence matcher
        cfg_rr = default_config(keynum=args.kpts, max_iter=1000, ET='yohoo', RD=True, RM=True)
        eng_rr = None
        if args.pipeline == 'rd_rm' or (args.rd_rm_steps > 0 and not args.no_secondary):
            rd, rm, rr_weights = rd_rm_nets(cfg_rr)
            eng_rr = RegistrationEngine(cfg_rr, gf, et, rd_net=rd, rm_net=rm)
            eng_rr.set_gemm_mode(args.gemm)
            if args.dtype == 'bf16':
                eng_rr.set_descriptor_dtype('bf16')
        if args.pipeline == 'rd_rm':
            eng, cfg = eng_rr, cfg_rr

    scenes, my_plan, totals = build_workload(args, rank, world, device=device, exchange=not args.no_exchange)   # inputs resident in HBM before timing
    transfers = totals['transfers']
    if forced and world == 1 and args.workload != 'chunk':
        transfers = D.self_transfers(my_plan, {s: scenes[s][3] for s in scenes}, rank=0, n_clouds=4)
    moved = {}                                                               # eqv bytes this rank sent / received, all steps
    seeds = {s: [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in scenes[s][3]] for s in scenes}
    kitchen = synth.THREEDMATCH_SCENES[0]
    kitchen_whole = any(p == (kitchen, 0, synth.THREEDMATCH_PAIRS[0]) for p in my_plan) and not any(t[0] == kitchen for t in transfers)
    kitchen_ms = []
    scene_index = {s: i for i, s in enumerate(synth.THREEDMATCH_SCENES + ['chunk'])}

    def scene_inputs(s):
        feats, keys, _, pairs = scenes[s]
        return feats, keys, pairs, seeds[s]

    def step(only=None, engine=None, **kw):
        """One pass of this rank's share + the step's single collective -> (this rank's rows [(scene, PairResult)], gathered table or None)."""
        pieces = [p for p in my_plan if only is None or p[0] == only]
        with D.watchdog(step_timeout if dist is not None else 0.0, 'one step (exchange, kernels, gather)'):
            done = D.run_plan(eng if engine is None else engine, pieces, scene_inputs, transfers if only is None else [], rank, stats=moved, **kw)
            rows = [(s, r) for s, _, _, res in done for r in res]
            table = None
            if only is None:                                                  # the single result-table collective (RCCL over xGMI)
                local_tab = np.concatenate([D.pack_rows(scene_index[s], res) for s, _, _, res in done] or [np.zeros((0, D.ROW))], 0)
                table = D.gather_table(local_tab, device=coll_dev, counts=totals['rows_per_rank']) if dist is not None else local_tab
        return rows, table

    def bracket(n, **kw):
        """n timed steps with the barrier + synchronise bracket on both sides; MAX over ranks."""
        barrier()
        sync()
        t1 = time.perf_counter()
        out = None
        for _ in range(n):
            out = step(**kw)
        sync()
        barrier()
        d = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([d], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d, out[0], out[1]

    barrier()            # the group's first call involves every rank (the warm-up's send/recv batch involves only those that cut a scene)
    for _ in range(args.warmup):
        step()

    if hip is not None:
        hip.PROFILE = []                                                     # per-launch HIP events of the group-conv GEMMs
        hip.profile_enable(True)                                             # library-side events: distance-matrix, RANSAC-scoring, Des2R kernels
    moved.clear()
    dt, rows, table = bracket(args.steps)
    moved_timed = dict(moved)
    prof = []
    if hip is not None:
        prof = hip.PROFILE; hip.PROFILE = None
        mm_ms, mm_n = hip.profile_read('mm_tile'); rs_ms, rs_n = hip.profile_read('ransac_score')
        d2_ms, d2_n = hip.profile_read('des2r'); ft_ms, ft_n = hip.profile_read('ft_nonlin')
        hip.profile_enable(False)
    my_pairs = sum(b - a for _, a, b in my_plan)
    Ms = np.array([r.n_match for _, r in rows], np.float64)
    Hs = np.minimum(Ms, 1000.0)

    # ---- the reference's per-pair work on the SAME workload: the local transform of every correspondence (test/estimator.py:308-367) ----
    sec = {}
    all_value = None
    metrics = None
    if not args.no_secondary:
        step(all_local_transforms=True)
        d_all, rows_all, table_all = bracket(args.all_steps, all_local_transforms=True, keep_matches=True)
        all_value = totals['pairs'] * args.all_steps / d_all
        same = all(np.array_equal(a.trans, b.trans, equal_nan=True) and a.recalltime == b.recalltime for (_, a), (_, b) in zip(rows, rows_all))
        if dist is not None:
            ok = torch.tensor([1 if same else 0], dtype=torch.int64, device=coll_dev); dist.all_reduce(ok, op=dist.ReduceOp.MIN); same = bool(ok.item())
        sec['results_identical_to_all_local_transforms'] = bool(same)
        # ---- FMR / IR / RR over ALL ranks (outside the timed regions; the evaluator's own metric code, test/evaluator.py:50-129) ----
        from roreg_amd.run_distributed import inlier_ratio, scene_metrics
        keys_host = {}
        for s, r in rows_all:
            for i in (int(r.id0), int(r.id1)):
                if (s, i) not in keys_host:
                    keys_host[(s, i)] = scenes[s][1][i].cpu().numpy()
            r.ir = inlier_ratio(cfg, r, keys_host[(s, int(r.id0))], keys_host[(s, int(r.id1))], synth.pose_transform(scenes[s][2], r.id0, r.id1))
        by_scene = {}
        for s, r in rows_all:
            by_scene.setdefault(s, []).append(r)
        local_tab = np.concatenate([D.pack_rows(scene_index[s], rs) for s, rs in by_scene.items()] or [np.zeros((0, D.ROW))], 0)
        full_tab = D.gather_table(local_tab, device=coll_dev) if dist is not None else local_tab
        if rank == 0 and args.workload != 'chunk':
            names = synth.THREEDMATCH_SCENES
            per = {}
            for row in D.unpack_rows(full_tab):
                per.setdefault(names[row['scene']], []).append(row)
            vals = []
            for s in sorted(per, key=names.index):
                # ground truth needs only the scene's poses: (group element, translation) per cloud from the scene's generator seed
                poses = scenes[s][2] if s in scenes else synth.scene_poses(500 + names.index(s), synth.THREEDMATCH_CLOUDS[names.index(s)])
                order = {(str(a), str(b)): q for q, (a, b) in enumerate(
                    synth.scene_pair_list(synth.THREEDMATCH_CLOUDS[names.index(s)], synth.THREEDMATCH_PAIRS[names.index(s)], 900 + names.index(s),
                                          locality=8.0 if args.pair_lists == 'banded' else None))}
                rs = sorted(per[s], key=lambda r: order[(r['id0'], r['id1'])])
                vals.append(scene_metrics(cfg, rs, lambda a, b: synth.pose_transform(poses, a, b)))
            v = np.array(vals)
            metrics = {'feature_matching_recall': float(v[:, 0].mean()), 'inlier_ratio': float(v[:, 1].mean()),
                       'registration_recall_pointdsc': float(v[:, 2].mean()), 'rotation_error_deg': float(np.nanmean(v[:, 3])),
                       'translation_error_m': float(np.nanmean(v[:, 4])), 'pairs': int(full_tab.shape[0]), 'scenes': len(vals),
                       'note': 'synthetic scenes (tau_1 = 0.05, tau_2 = 0.1 m; RR = RRE < 15 deg and RTE < 0.3 m, scene means averaged like '
                               'test/evaluator.py:111-129); RR(predator) needs the benchmark\'s gt.info covariances, which synthetic scenes do not have'}

    # ---- the strictly 24-bit matrix-core mode on the SAME full workload (the default f16x2 is 22 bits under a per-keypoint bound) ----
    bf16x3_value = None
    bf16x3_all_value = None
    if not args.no_secondary and hip is not None and args.gemm == 'f16x2' and args.bf16x3_steps > 0:
        eng.set_gemm_mode('bf16x3')
        step()
        d_b, _, _ = bracket(args.bf16x3_steps)
        bf16x3_value = totals['pairs'] * args.bf16x3_steps / d_b
        step(all_local_transforms=True)
        d_ba, _, _ = bracket(args.bf16x3_steps, all_local_transforms=True)
        bf16x3_all_value = totals['pairs'] * args.bf16x3_steps / d_ba
        eng.set_gemm_mode(args.gemm)


    # ---- RoReg's own pipeline -- `Test.py --RD --RM --ET yohoo --keynum 5000` (README.md:149,175; SURVEY 3.1: THE hot path) -- on the SAME full workload ----
    rr_full = None
    if hip is not None and eng_rr is not None and args.pipeline != 'rd_rm' and args.rd_rm_steps > 0:
        rr_full = rd_rm_full(args, totals, eng_rr, step, bracket, scenes, rr_weights, dist, coll_dev)

    # ---- secondary figures (outside the headline's timed region; kitchen scene only, so the default run stays short) ----
    sec_scene = kitchen if kitchen in scenes else sorted(scenes)[0]
    if not args.no_secondary and kitchen_whole:                               # BASELINE configs[1]: the kitchen scene alone on this GPU
        step(only=kitchen)
        for _ in range(3):
            sync(); tk = time.perf_counter()
            step(only=kitchen)
            sync(); kitchen_ms.append(1e3 * (time.perf_counter() - tk))
    if not args.no_secondary and hip is not None:
        n_sec = 1
        d_def, rows_def, _ = bracket(n_sec, only=sec_scene)
        n_sec_pairs = len(rows_def)
        sec['secondary_scene'] = sec_scene
        other = {}
        for mode in [m for m in ('f16x2', 'bf16x3', 'f32') if m != args.gemm]:
            eng.set_gemm_mode(mode)
            step(only=sec_scene)
            hip.PROFILE = []
            d_o, rows_o, _ = bracket(n_sec, only=sec_scene)
            pr = hip.PROFILE; hip.PROFILE = None
            ach, ms_o, nl_o, alg_o = gemm_roofline(pr, TAG_OF[mode])
            diffs = [float(np.abs(a.trans - b.trans).max()) for (_, a), (_, b) in zip(rows_def, rows_o) if np.isfinite(a.trans).all() and np.isfinite(b.trans).all()]
            other[mode] = {'dtype': DTYPE_OF[mode], 'value_on_secondary_scene': world * n_sec_pairs * n_sec / d_o if n_sec_pairs else None,
                           'ms_per_pass': 1e3 * d_o / n_sec, 'pairs_with_identical_transform_to_default': int(sum(1 for d in diffs if d == 0.0)),
                           'pairs_compared': len(diffs), 'max_abs_diff_of_transforms_vs_default': max(diffs or [0.0]),
                           'roofline': roofline_obj(mode, ach, ms_o, nl_o, None, alg_o)}
        eng.set_gemm_mode(args.gemm)
        sec['value_default_on_secondary_scene'] = world * n_sec_pairs * n_sec / d_def if n_sec_pairs else None
        sec['other_gemm_modes'] = other
        eng.phase_ms = {}
        step(only=sec_scene)
        sec['phase_ms_one_synchronised_pass_of_secondary_scene'] = {k: round(v, 2) for k, v in eng.phase_ms.items()}
        eng.phase_ms = None
        if args.gemm == 'f16x2':
            sec['f16x2_scale_headroom_bits'] = headroom_bits(eng, scenes[sec_scene][0][0])
        if world == 1 and not args.no_dropin:
            try:
                sec['dropin_leg'] = dropin_leg(args, gf_sd, et_sd)
            except OSError as e:                                              # (no room for the scene's files in the temporary directory)
                sec['dropin_leg'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and kitchen in scenes:
            sec['yohoc_leg'] = yohoc_leg(args, gf, scenes, seeds, kitchen)
            sec['yohoc_leg_pairs_per_s'] = sec['yohoc_leg']['yohoc']['pairs_per_s']
        if world == 1:
            sec['rd_rm_leg'] = rd_rm_leg(args, cfg, gf, et)
            sec['rd_rm_leg_pairs_per_s'] = sec['rd_rm_leg'].get('fp32', {}).get('pairs_per_s')
            sec['rd_rm_leg_pairs_per_s_bf16'] = sec['rd_rm_leg'].get('bf16', {}).get('pairs_per_s')
            sec['rd_rm_leg_pairs_per_s_matrix_core_layers'] = sec['rd_rm_leg'].get('fp32_matrix_core_layers', {}).get('pairs_per_s')
            sec['rd_rm_leg_sinkhorn_ms_per_pair'] = (sec['rd_rm_leg'].get('fp32', {}).get('roofline_sinkhorn') or {}).get('ms_per_pair')
            # the same chain at `--keynum 5000` (SURVEY 3.1's hot path: Match_ot at m = n = 5000)
            sec['rd_rm_leg_k5000'] = rd_rm_leg(args, cfg, gf, et, keynum=5000, n_pairs=52, variants=('fp32', 'fp32_matrix_core_layers'))
            sec['rd_rm_leg_k5000_pairs_per_s'] = sec['rd_rm_leg_k5000'].get('fp32', {}).get('pairs_per_s')
            sec['rd_rm_leg_k5000_pairs_per_s_matrix_core_layers'] = sec['rd_rm_leg_k5000'].get('fp32_matrix_core_layers', {}).get('pairs_per_s')
            sec['rd_rm_leg_k5000_sinkhorn_ms_per_pair'] = (sec['rd_rm_leg_k5000'].get('fp32', {}).get('roofline_sinkhorn') or {}).get('ms_per_pair')

    # ---- roofline of the dominant kernel: the irrep-domain GEMMs of the two big GF layers (256->512, 512->256) ----
    if hip is not None:
        achieved, ms, n_launch, alg_bytes = gemm_roofline(prof, TAG_OF[args.gemm])
        traffic = measured_traffic(args, world)

    eqv_moved = float(moved_timed.get('eqv_bytes_sent', 0))
    if dist is not None:
        tt = torch.tensor([eqv_moved], dtype=torch.float64, device=coll_dev); dist.all_reduce(tt); eqv_moved = float(tt.item())
    eqv_moved /= max(args.steps, 1)

    if rank == 0:
        total_pairs = totals['pairs']
        value = total_pairs * args.steps / dt
        wl = {'3dmatch-full': f"full 3DMatch test shape, synthetic: 8 scenes, {totals['clouds']} clouds x {args.kpts} kpts, {total_pairs} pairs "
                              f"(mutual matcher + yohoo estimator, max_iter=1000, 60-rot group feats); pairs sharded over the ranks by scene",
              'kitchen': f"3DMatch-kitchen-like scene: {totals['clouds']} clouds x {args.kpts} kpts, {total_pairs} pairs (mutual + yohoo, max_iter=1000)",
              'chunk': f"scene chunk per GPU: 16 clouds x {args.kpts} kpts, 60 pairs (mutual + yohoo, max_iter=1000)"}[args.workload]
        if args.pipeline == 'rd_rm':
            wl = wl.replace('mutual matcher + yohoo estimator', 'detector + NMS + rotation-coherence matcher + yohoo estimator: --RD --RM').replace('mutual + yohoo', '--RD --RM + yohoo')
        out = {
            'metric': 'pair-registrations/sec', 'value': value, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': totals['scaling'],
            'vs_baseline': None, 'dtype': ('stub' if STUB else DTYPE_OF[args.gemm] + ('' if args.dtype == 'fp32' else '; group features stored as bfloat16')),
            'data': 'stub engine (host-side test of the launcher and the multi-rank control flow; no kernels ran)' if STUB else 'synthetic',
            # the like-for-like figures next to `value` (which evaluates only the <= 1000 local transforms one-shot RANSAC draws; identical results):
            # the reference's per-pair work and inter-stage file contract -- the local transform of EVERY correspondence -- in the default
            # matrix-core mode and in the strictly 24-bit one
            'value_contract_complete': all_value,
            'value_contract_complete_bf16x3': bf16x3_all_value,
            'value_all_local_transforms': all_value,
            'accuracy': metrics,
            'value_bf16x3': bf16x3_value,
            **({k: v for k, v in rr_full.items() if k != 'rd_rm_k5000'} if rr_full else {}),
            'config': {'workload': wl, 'pairs_per_step': total_pairs, 'clouds_per_step': totals['clouds'], 'parallelism': f'pairs-sharded x{world}',
                       'descriptor_dtype': args.dtype,
                       # scalars repeated here because a record that keeps only the contract's keys keeps `config`'s scalars:
                       'value_all_local_transforms': all_value,            # the reference's per-pair work (every correspondence's local transform)
                       'value_bf16x3': bf16x3_value,                        # same workload, strictly 24-bit operands (3 x bf16) in the matrix cores
                       'value_contract_complete': all_value, 'value_contract_complete_bf16x3': bf16x3_all_value,
                       'value_rd_rm_k5000': None if not rr_full else rr_full['value_rd_rm_k5000'],
                       'value_rd_rm_k5000_contract_complete': None if not rr_full else rr_full['value_rd_rm_k5000_contract_complete'],
                       'value_rd_rm_k5000_all_sinkhorn_iterations': None if not rr_full else rr_full['value_rd_rm_k5000_all_sinkhorn_iterations'],
                       'rd_rm_k5000': None if not rr_full else rr_full['rd_rm_k5000'],
                       'fmr': None if metrics is None else metrics['feature_matching_recall'],
                       'ir': None if metrics is None else metrics['inlier_ratio'],
                       'rr': None if metrics is None else metrics['registration_recall_pointdsc'],
                       'eqv_bytes_moved_per_step': eqv_moved,              # extractor outputs sent rank to rank (sum over ranks), timed steps
                       'forced_collectives': bool(forced and world == 1),
                       'pair_lists': ('chain (i, i+1) + pairs (i, j) drawn with probability ~ exp(-|i-j|/8): scan-sequence-like locality' if args.pair_lists == 'banded'
                                      else 'chain (i, i+1) + pairs drawn uniformly over all cloud pairs') if args.workload != 'chunk' else None,
                       'rank0_pairs_per_step': my_pairs, 'shard_plan': totals.get('plan'),
                       'cloud_extractions_per_rank': totals.get('extractions_per_rank'),
                       'eqv_transfers_per_step': len(transfers), 'result_table_bytes_gathered_per_step': (total_pairs * D.ROW * 8 if dist is not None else 0),
                       'eqv_exchange': ('point-to-point send/recv of cut scenes\' extractor outputs (38.4 MB each), one grouped '
                                                                                   'launch per step' if transfers else None),
                       'mean_matches_rank0': float(np.mean(Ms)) if len(Ms) else None,
                       'local_transforms': 'value: only the <=1000 hypotheses one-shot RANSAC draws per pair (results identical to evaluating all M, checked: '
                                           'results_identical_to_all_local_transforms); value_all_local_transforms: every correspondence, as the reference '
                                           f'computes them (same workload, {args.all_steps} timed steps)',
                       'kitchen_scene': ({'clouds': synth.THREEDMATCH_CLOUDS[0], 'pairs': synth.THREEDMATCH_PAIRS[0], 'ms_per_pass': float(np.mean(kitchen_ms)),
                                          'pairs_per_s': synth.THREEDMATCH_PAIRS[0] / (np.mean(kitchen_ms) * 1e-3),
                                          'note': 'BASELINE configs[1]: the kitchen scene alone, three synchronised passes outside the headline region'} if kitchen_ms else None),
                       **sec},
        }
        if dist is not None:
            out['config']['devices'] = 'every rank on GPU 0 (ROREG_BENCH_SHARED_GPU)' if os.environ.get('ROREG_BENCH_SHARED_GPU') else ('cpu (stub)' if STUB else list(range(world)))
            out['config']['backend'] = args.backend
        if hip is not None:
            mm_flop = float(np.sum(2.0 * args.kpts * args.kpts * 32)) * len(rows) * args.steps            # algorithmic: one N x N x 32 product per pair
            rs_flop = float(np.sum(Hs * Ms * 27.0)) * args.steps
            rs_bytes = float(np.sum(Ms * 56.0 + Hs * 96.0)) * args.steps
            d2_bytes = float(np.sum(Hs * 15360.0)) * args.steps
            out['roofline'] = roofline_obj(args.gemm, achieved, ms, n_launch, traffic, alg_bytes)
            # the kernels north_star names, measured with HIP events on the launch stream in the same timed steps (rank 0)
            out['roofline_distance_gemm'] = None if not mm_n else {
                'kernel': 'mm_strip_kernel<false> + mm_strip_kernel<true> (5000 x 5000 x 32 descriptor distances of every pair of a scene, both search directions; '
                          'fp16 x 2 split MFMA bound under per-row scales + exact re-check; one workgroup per 128-row strip, column tiles streamed by LDS-DMA)',
                'bound': 'mfma', 'unit': 'TFLOP/s', 'launch_pairs': mm_n, 'avg_ms': mm_ms / mm_n,
                'algorithmic_tflops': mm_flop / (mm_ms * 1e-3) / 1e12, 'achieved': 6.0 * mm_flop / (mm_ms * 1e-3) / 1e12, 'peak': PEAK_BF16_MFMA_TFLOPS,
                'frac': 6.0 * mm_flop / (mm_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                'note': 'executed = 6 x algorithmic: two products (S.T^T in pass A for the row AND column minima, again in pass B for the candidate test) x '
                        'three fp16 MFMAs per f32-accurate product.  With K = 32 the matrix cores are a small part of the kernel: the rest is the operand '
                        'split and the min / candidate search over every one of the tile elements on the vector ALUs'}
            out['roofline_ransac'] = None if not rs_n else {
                'kernel': 'ransac_score_batch_kernel (one wave per hypothesis over the pair\'s M correspondences, fp64, no FMA contraction: bit-exact masks)',
                'bound': 'fp64-valu', 'unit': 'TFLOP/s', 'launches': rs_n, 'avg_ms': rs_ms / rs_n, 'achieved': rs_flop / (rs_ms * 1e-3) / 1e12,
                'peak': PEAK_F64_VALU_TFLOPS, 'frac': rs_flop / (rs_ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS,
                'hbm': {'algorithmic_GBps': rs_bytes / (rs_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'frac': rs_bytes / (rs_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                        'note': 'M*56 + H*96 B per pair: the correspondences are re-read by every hypothesis from L2, the stage is not HBM-bound'}}
            out['roofline_des2r'] = None if not d2_n else {
                'kernel': 'des2r_irrep_batch_kernel (irrep-domain bound + exact re-check of near ties; M*15,360 algorithmic bytes)', 'bound': 'hbm', 'unit': 'GB/s', 'launches': d2_n, 'avg_ms': d2_ms / d2_n,
                'achieved': d2_bytes / (d2_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'frac': d2_bytes / (d2_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
            out['transforms'] = None if not ft_n else {'kernel': 'ft_nonlin_kernel (all variants)', 'launches': ft_n, 'ms_per_step': ft_ms / args.steps}
        if hip is not None and not args.no_cpu_baseline and world == 1:
            s0 = sorted(scenes)[0]
            f0 = scenes[s0][0][0].float().cpu().numpy(); f1 = scenes[s0][0][1].float().cpu().numpy()
            M = float(np.mean(Ms))
            threads = torch.get_num_threads()
            t_t = cpu_baseline_torch((gf_sd, et_sd), f0, f1)
            v_all, v_drawn, comp = scale_baseline(t_t, M, totals['clouds'], total_pairs)
            out['cpu_baseline'] = {'value': v_all, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'sampled': True,
                                   'sample': 'SAMPLED, not a timed step (per-unit costs on bounded samples, scaled to the workload): ' 'oracle/ref_torch.py (the reference\'s own torch CPU operators, torch.set_num_threads = all host threads) on this host: '
                                             'GF on one bs_GF batch of 1250 kpts, mutual NN on the full 5000 x 5000 pair, Des2R on 1000 and ET on one bs_ET batch of '
                                             '1000 correspondences, RANSAC scoring (numpy, as in the reference) on 200 hypotheses x 3000; scaled to the step workload '
                                             'with the local transform of EVERY correspondence, as the reference computes it (pair with value_all_local_transforms)',
                                   'value_if_only_drawn_hypotheses': v_drawn, 'components_s': comp}
            t_n = cpu_baseline_numpy(({k: v.numpy() for k, v in gf_sd.items()}, {k: v.numpy() for k, v in et_sd.items()}), f0, f1)
            v_all, v_drawn, comp = scale_baseline(t_n, M, totals['clouds'], total_pairs)
            out['cpu_baseline_numpy_oracle'] = {'value': v_all, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'sampled': True,
                                                'sample': 'oracle/ref_numpy.py (the parity oracle; BLAS-threaded matmuls, single-threaded gathers): GF on 512 kpts, '
                                                          'mutual on 2500 x 2500 (scaled N^2), Des2R on 512, ET on 384 correspondences, RANSAC scoring on 200 x 3000',
                                                'value_if_only_drawn_hypotheses': v_drawn, 'components_s': comp}
        record_out.write(json.dumps(out) + '\n'); record_out.flush()
    if dist is not None:
        barrier()
        dist.destroy_process_group()


def headroom_bits(eng, feats):
    """How far the fp16 x 2 mode's per-keypoint block-scale BOUND sits above the keypoint's true coefficient maximum, per GEMM input of the
    extractor, on one cloud: log2(bound / max|coef|) min / mean / max -- the bits of the 22 the conservative bound gives away."""
    net = eng.gf.PartI_net
    try:
        rep = net._fourier.scale_headroom(feats.float())
    except Exception as e:                                                  # diagnostics only
        return {'error': f'{type(e).__name__}: {e}'}
    return rep


def sinkhorn_roofline(work, sk_ms, sk_n, fused):
    """MFMA roofline of the recomputing Sinkhorn iterations (csrc/ot_flash.hip) from the library's HIP-event brackets: executed fp16 MFMA flop per
    coupling-matrix element and iteration (seven K = 16 MFMAs per 32 x 32 tile = 224; twice that when the scores are recomputed in two passes)."""
    cells = work.get('sinkhorn_cells', 0.0)
    per_cell = 224.0 if fused else 448.0
    tf = cells * per_cell / (sk_ms * 1e-3) / 1e12
    return {'kernel': 'of_iter_kernel + of_update_cols_kernel (one recomputation per iteration)' if fused else 'of_pass_kernel x 2 + of_update kernels (two recomputing passes per iteration)',
            'bound': 'mfma', 'unit': 'TFLOP/s', 'launch_groups': sk_n, 'avg_ms': sk_ms / max(sk_n, 1), 'ms_per_pair': sk_ms / max(work.get('sinkhorn_pairs', 1), 1),
            'achieved': tf, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': tf / PEAK_BF16_MFMA_TFLOPS, 'executed_mfma_flop_per_element_and_iteration': per_cell,
            'traffic': None, 'hbm_bytes_not_read': work.get('sinkhorn_bytes', 0.0)}


def rd_rm_full(args, totals, eng_rr, step, bracket, scenes, weights, dist, coll_dev):
    """RoReg's own pipeline on the full benchmark shape: detector -> NMS -> rotation-coherence matcher (Match_ot at m = n = keynum) -> yohoo on the top
    `match_n` matches, through the same run_plan / run_scenes path as the headline.  -> dict of top-level keys for the record.
    Timed twice: as the library runs it by default (a pair's Sinkhorn iterations stop at the float32 fixed point, include/roreg_hip.h v6), and with
    every pair run through all 100 iterations like the reference's loop (network/rot_coh_match.py:289-292) -- the second also prices the Sinkhorn
    kernel against the MFMA peak (executed flops are then known exactly)."""
    from roreg_amd import hip, synth
    from roreg_amd.utils.r_eval import compute_R_diff
    n = args.rd_rm_steps
    step(engine=eng_rr)                                                          # warm-up at the timed depth
    hip.profile_enable(True); hip.WORK = {}; hip.sinkhorn_iteration_stats()
    d, rows, _ = bracket(n, engine=eng_rr, keep_matches=True)
    sk_ms, sk_n = hip.profile_read('sinkhorn'); tk_ms, tk_n = hip.profile_read('topk_dot'); ft_ms, ft_n = hip.profile_read('ft_nonlin')
    work = hip.WORK; hip.WORK = None; hip.profile_enable(False)
    it_run, it_pairs = hip.sinkhorn_iteration_stats()
    value = totals['pairs'] * n / d
    # the same with every pair run through all the iterations
    with hip.sinkhorn_early_exit(False):
        step(engine=eng_rr)
        hip.profile_enable(True); hip.WORK = {}
        d_fix, rows_fix, _ = bracket(1, engine=eng_rr, keep_matches=True)
        skf_ms, skf_n = hip.profile_read('sinkhorn')
        work_fix = hip.WORK; hip.WORK = None; hip.profile_enable(False)
    # what stopping at the fixed point changes: the match lists (the index outputs), the matching scores, and -- where the pair registers at all on
    # this synthetic data (most do not under the shipped RM weights, and a failed registration is a function of the scores' last bits) -- the transform
    same_m = 0; dsc = []; dT_ok = []; n_ok = n_ok_a = n_ok_b = 0
    for (sc, a), (_, b) in zip(rows, rows_fix):
        same = a.n_match == b.n_match and torch.equal(a.matches, b.matches)
        same_m += int(same)
        if same and a.scores is not None and len(a.scores):
            dsc.append(float(np.abs(a.scores - b.scores).max()))
        gt = synth.pose_transform(scenes[sc][2], a.id0, a.id1)
        good = lambda r: bool(np.isfinite(r.trans).all() and compute_R_diff(r.trans[:3, :3], gt[:3, :3]) < 15 and np.sqrt(np.sum(np.square(r.trans[:3, 3] - gt[:3, 3]))) < 0.3)
        n_ok_a += int(good(a)); n_ok_b += int(good(b))
        if good(a) or good(b):
            n_ok += 1
            dT_ok.append(float(np.abs(a.trans - b.trans).max()) if good(a) and good(b) else float('inf'))
    step(engine=eng_rr, all_local_transforms=True)
    d_all, rows_all, _ = bracket(1, engine=eng_rr, all_local_transforms=True)
    same = all(np.array_equal(a.trans, b.trans, equal_nan=True) and a.recalltime == b.recalltime for (_, a), (_, b) in zip(rows, rows_all))
    eng_rr.phase_ms = {}                                                          # one synchronised (un-pipelined) pass: where the time goes, by stage
    step(engine=eng_rr)
    phases = {k: round(v, 2) for k, v in eng_rr.phase_ms.items()}; eng_rr.phase_ms = None
    ok = []
    for s, r in rows:
        gt = synth.pose_transform(scenes[s][2], r.id0, r.id1)
        ok.append(bool(np.isfinite(r.trans).all() and compute_R_diff(r.trans[:3, :3], gt[:3, :3]) < 15 and np.sqrt(np.sum(np.square(r.trans[:3, 3] - gt[:3, 3]))) < 0.3))
    out = {'value_rd_rm_k5000': value, 'value_rd_rm_k5000_contract_complete': totals['pairs'] / d_all,
           'value_rd_rm_k5000_all_sinkhorn_iterations': totals['pairs'] / d_fix,
           'rd_rm_k5000': {'workload': f"the headline's workload ({totals['clouds']} clouds x {args.kpts} kpts, {totals['pairs']} pairs) through --RD --RM --ET yohoo --keynum {eng_rr.cfg.keynum} "
                                       f"--match_n {eng_rr.cfg.match_n}: detector, NMS sampling, Match_ot at m = n = {eng_rr.cfg.keynum}, one-shot estimator on the top matches",
                           'weights': weights + '; seeded GF / ET weights', 'steps': n, 'ms_per_step': 1e3 * d / n, 'ms_per_step_contract_complete': 1e3 * d_all,
                           'ms_per_step_all_sinkhorn_iterations': 1e3 * d_fix,
                           'sinkhorn': {'iterations_asked': 100, 'iterations_run_mean': it_run / max(it_pairs, 1), 'pairs': it_pairs // max(n, 1),
                                        'rule': 'a pair stops at the fixed point of the float32 iteration: once an iteration\'s largest step of a potential is <= max(2^-22 |u|, 2^-20) in log2 '
                                                'units (2 .. 4 float32 ulps), or <= 8 such units and not smaller than the previous iteration\'s: the noise floor '
                                                'of the recomputed scores; roreg_sinkhorn_early_exit(0) runs all of them',
                                        'ms_per_pair': sk_ms / max(work.get('sinkhorn_pairs', 1), 1) if sk_n else None,
                                        'ms_per_pair_all_iterations': skf_ms / max(work_fix.get('sinkhorn_pairs', 1), 1) if skf_n else None,
                                        'pairs_with_identical_match_lists_to_all_iterations': same_m, 'pairs_compared': len(rows),
                                        'max_abs_diff_of_matching_scores_on_those': max(dsc or [0.0]),
                                        'pairs_registered': n_ok_a, 'pairs_registered_all_iterations': n_ok_b, 'pairs_registered_in_either_run': n_ok, 'of_those_registered_in_both': int(sum(1 for x in dT_ok if np.isfinite(x))),
                                        'max_abs_diff_of_transforms_on_those': max([x for x in dT_ok if np.isfinite(x)] or [0.0])},
                           'results_identical_to_all_local_transforms': bool(same), 'mean_matches_rank0': float(np.mean([r.n_match for _, r in rows])) if rows else None,
                           'registration_recall_pointdsc_rank0': float(np.mean(ok)) if ok else None,
                           'stage_ms_one_synchronised_pass_rank0': phases,
                           'transforms_ms_per_step': ft_ms / n if ft_n else None,
                           'topk_dot_ms_per_step': tk_ms / n if tk_n else None,
                           'sinkhorn_ms_per_step': sk_ms / n if sk_n else None}}
    if skf_n and work_fix.get('sinkhorn_recompute'):
        fused = os.environ.get('ROREG_OT_FUSED', '1') != '0' and (eng_rr.cfg.keynum <= 2559 or bool(getattr(hip, 'OT_COOP', False)))
        out['roofline_rd_rm'] = dict(sinkhorn_roofline(work_fix, skf_ms, skf_n, fused), measured_on='the pass with every pair run through all 100 iterations (executed flops known exactly)')
    return out


def dropin_leg(args, gf_sd, et_sd, n_clouds=60, n_pairs=449):
    """The boundary north_star names -- `Test.py` unchanged, i.e. `yoho_evaluator(cfg).process_scene(dataset)` (Test.py:20-23, test/evaluator.py:39-48)
    -- on ONE kitchen-shaped scene (60 clouds x 5000 kpts, 449 pairs, mutual matcher + yohoo) laid out on disk the way testset.py leaves it
    ({cache}/{scene}/FCGF_Input_Group_feature/{pc}.npy, checkpoints under --model_fn), timed three ways:
      stages  : the file-coupled chain of stage.run() calls (ROREG_EVALUATOR=stages: every stage reads its inputs from .npy and writes its outputs),
      engine  : what process_scene does by default -- the device-resident engine + StageFileWriter: the same files, written asynchronously,
      no_files: the engine alone on inputs already resident in HBM, every correspondence's local transform evaluated (the same per-pair work), no file.
    Inputs are page-cached (just written); every output file of `engine` is compared with `stages` (.npy byte for byte, .npz by content)."""
    import filecmp
    import glob
    import shutil
    import tempfile
    from roreg_amd import hip, synth
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    from roreg_amd.test import _cache
    from roreg_amd.test.evaluator import yoho_evaluator
    root = tempfile.mkdtemp(prefix='roreg_dropin_', dir=os.environ.get('ROREG_BENCH_TMP'))
    try:
        feats, keys, poses = synth.make_scene_device(777, n_clouds, args.kpts, OVERLAP)
        pairs = synth.scene_pair_list(n_clouds, n_pairs, 901, locality=8.0)
        ds = synth.SynthScene('synth/kitchen', [k.cpu().numpy() for k in keys], None, poses, pairs)
        inputs = f'{root}/inputs/FCGF_Input_Group_feature'
        os.makedirs(inputs)
        for i, f in enumerate(feats):
            np.save(f'{inputs}/{i}.npy', f.cpu().numpy())
        for kind, sd in (('GF', gf_sd), ('ET', et_sd)):
            os.makedirs(f'{root}/ckpt/{kind}')
            torch.save({'best_para': 0, 'network_state_dict': sd}, f'{root}/ckpt/{kind}/model_best.pth')
        out = {'workload': f'one scene on disk: {n_clouds} clouds x {args.kpts} kpts, {n_pairs} pairs, mutual matcher + yohoo (Test.py --ET yohoo --keynum {args.kpts}), '
                           f'{n_clouds * args.kpts * 1920 * 4 / 1e9:.1f} GB of input features, page-cached', 'tmp': os.path.dirname(root)}
        before = os.environ.get('ROREG_EVALUATOR')
        for route in ('stages', 'engine'):
            os.environ['ROREG_EVALUATOR'] = route
            runs = []
            phases = split = None
            cfg = default_config(output_cache_fn=f'{root}/cache_{route}_0', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None, keynum=args.kpts, max_iter=1000, ET='yohoo')
            _cache.clear()
            ev = yoho_evaluator(cfg)                              # ONE evaluator per route, like a benchmark run: its scenes share the loaded networks
            if route == 'engine':
                ev._engine().set_gemm_mode(args.gemm)
            else:
                hip.GEMM_MODE = args.gemm
            timed = (1, 2, 3) if route == 'engine' else (1,)      # the scene into fresh cache directories: the runs AFTER the first are reported (a benchmark is 8 scenes;
            for rep in range(timed[-1] + (2 if route == 'engine' else 1)):      # the first one also pays for the weights' packing and the allocators' first pinned blocks);
                cache = f'{root}/cache_{route}' if rep == 1 else f'{root}/cache_{route}_{rep}'      # the engine route once more with synchronised stage marks (diagnostics only)
                cfg.output_cache_fn = cache
                os.makedirs(f'{cache}/{ds.name}')
                os.symlink(inputs, f'{cache}/{ds.name}/FCGF_Input_Group_feature')
                _cache.clear()
                if route == 'engine' and rep == timed[-1] + 1:
                    ev._engine().phase_ms = {}
                np.random.seed(5)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ev.process_scene(ds)
                torch.cuda.synchronize(); runs.append(time.perf_counter() - t0)
                if rep == 1:
                    split = getattr(ev, 'last_scene_seconds', None)
                if route == 'engine' and rep == timed[-1] + 1:
                    phases = {k: round(v, 1) for k, v in ev._engine().phase_ms.items()}
                    ev._engine().phase_ms = None
                if rep != 1:
                    shutil.rmtree(cache, ignore_errors=True)
            del ev
            mean = float(np.mean([runs[r] for r in timed]))
            out[route] = {'pairs_per_s': n_pairs / mean, 's_per_scene': mean, 's_each_scene_after_the_first': [round(runs[r], 4) for r in timed], 's_first_scene': runs[0]}
            if split:
                out[route]['seconds'] = {k: round(v, 3) for k, v in split.items()}
            if phases:
                out[route]['stage_ms_one_synchronised_run'] = phases
        # the same route under the estimator an unflagged Test.py runs (parses/parses_test.py:42: --ET yohoc): draws from the process-global generator
        # in pair order on the launching thread, 3-point Kabsch stacks on the host pool (files equal the stage chain's: tests/test_hip_pipeline.py)
        os.environ['ROREG_EVALUATOR'] = 'engine'

        def other_flags(tag, what, **flags):
            cfg = default_config(output_cache_fn=f'{root}/cache_{tag}_0', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None, keynum=args.kpts, max_iter=1000, **flags)
            ev = yoho_evaluator(cfg)
            ev._engine().set_gemm_mode(args.gemm)
            runs = []
            for rep in range(4):
                cache = f'{root}/cache_{tag}_{rep}'
                cfg.output_cache_fn = cache
                os.makedirs(f'{cache}/{ds.name}')
                os.symlink(inputs, f'{cache}/{ds.name}/FCGF_Input_Group_feature')
                _cache.clear()
                np.random.seed(5)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ev.process_scene(ds)
                torch.cuda.synchronize(); runs.append(time.perf_counter() - t0)
                shutil.rmtree(cache, ignore_errors=True)
            del ev
            return {'pairs_per_s': n_pairs / float(np.mean(runs[1:])), 's_per_scene': float(np.mean(runs[1:])), 's_each_scene_after_the_first': [round(x, 4) for x in runs[1:]],
                    's_first_scene': runs[0], 'what': what}
        out['engine_yohoc'] = other_flags('yohoc', 'Test.py without --ET (yohoc, parses_test.py:42), same scene, same file contract', ET='yohoc')
        # ... and under the README's own command (README.md:149,175: --RD --RM --ET yohoo --keynum 5000; the shipped RD / RM checkpoints when the fixtures hold them)
        rd, rm, rd_rm_weights = rd_rm_nets(default_config(keynum=args.kpts, max_iter=1000, ET='yohoo', RD=True, RM=True))
        for kind, net in (('RD', rd), ('RM', rm)):
            os.makedirs(f'{root}/ckpt/{kind}')
            torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{root}/ckpt/{kind}/model_best.pth')
        del rd, rm
        out['engine_rd_rm'] = other_flags('rd_rm', f'Test.py --RD --RM --ET yohoo --keynum {args.kpts} (README.md:149), same scene, the detector score files too; {rd_rm_weights}',
                                          ET='yohoo', RD=True, RM=True)
        if before is None:
            os.environ.pop('ROREG_EVALUATOR', None)
        else:
            os.environ['ROREG_EVALUATOR'] = before
        _cache.clear()
        a, b = f'{root}/cache_stages/{ds.name}', f'{root}/cache_engine/{ds.name}'
        rel = sorted(os.path.relpath(f, a) for f in glob.glob(f'{a}/**/*.npy', recursive=True) if 'Input_Group_feature' not in f)
        rel_b = sorted(os.path.relpath(f, b) for f in glob.glob(f'{b}/**/*.npy', recursive=True) if 'Input_Group_feature' not in f)
        differ = [r for r in rel if r not in rel_b or not filecmp.cmp(f'{a}/{r}', f'{b}/{r}', shallow=False)]
        res = sorted(os.path.relpath(f, a) for f in glob.glob(f'{a}/**/*.npz', recursive=True))
        res_differ = []
        for r in res:
            x, y = np.load(f'{a}/{r}'), np.load(f'{b}/{r}') if os.path.exists(f'{b}/{r}') else None
            if y is None or sorted(x.files) != sorted(y.files) or not all(np.array_equal(x[k], y[k], equal_nan=True) for k in x.files):
                res_differ.append(r)
        out['files'] = {'npy_files_compared_byte_for_byte': len(rel), 'npy_files_that_differ': len(differ) + len(set(rel_b) - set(rel)), 'result_npz_compared': len(res),
                        'result_npz_that_differ': len(res_differ), 'bytes_written_per_route': int(sum(os.path.getsize(f'{a}/{r}') for r in rel)),
                        'examples_of_differing_files': (differ + res_differ)[:4]}
        # what the temporary directory takes: the scene's extractor outputs (60 x 38.4 MB) through np.save on four threads, from memory
        from concurrent.futures import ThreadPoolExecutor
        blob = np.zeros((args.kpts, 32, 60), np.float32)
        os.makedirs(f'{root}/io_probe')
        t0 = time.perf_counter()
        with ThreadPoolExecutor(4) as pool:
            list(pool.map(lambda i: np.save(f'{root}/io_probe/{i}.npy', blob), range(n_clouds)))
        dt_io = time.perf_counter() - t0
        out['files']['np_save_of_the_extractor_outputs_alone_s'] = round(dt_io, 3)
        out['files']['np_save_GB_per_s'] = round(n_clouds * blob.nbytes / dt_io / 1e9, 2)
        shutil.rmtree(f'{root}/io_probe', ignore_errors=True)
        # the engine alone, inputs resident in HBM, the same per-pair work (every correspondence's local transform), no files
        cfg = default_config(keynum=args.kpts, max_iter=1000, ET='yohoo')
        gf = name2network['GF_test'](cfg); gf.load_state_dict(gf_sd)
        et = name2network['ET_test'](cfg); et.load_state_dict(et_sd)
        eng = RegistrationEngine(cfg, gf, et); eng.set_gemm_mode(args.gemm)
        np.random.seed(5)
        eng.run_scene(feats, keys, ds.pair_ids, all_local_transforms=True)
        dts = []
        for _ in range(3):
            np.random.seed(5)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run_scene(feats, keys, ds.pair_ids, all_local_transforms=True)
            torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
        dt = float(np.mean(dts))
        out['no_files'] = {'pairs_per_s': n_pairs / dt, 's_per_scene': dt, 's_each_scene_after_the_first': [round(x, 4) for x in dts]}
        eng.phase_ms = {}
        np.random.seed(5)
        eng.run_scene(feats, keys, ds.pair_ids, all_local_transforms=True)
        out['no_files']['stage_ms_one_synchronised_run'] = {k: round(v, 1) for k, v in eng.phase_ms.items()}
        eng.phase_ms = None
        out['engine_over_stages'] = out['engine']['pairs_per_s'] / out['stages']['pairs_per_s']
        out['engine_over_no_files'] = out['engine']['pairs_per_s'] / out['no_files']['pairs_per_s']
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def yohoc_leg(args, gf, scenes, seeds, kitchen, n_rep=4):
    """The estimator an unflagged Test.py runs (parses/parses_test.py:42: --ET yohoc, test/estimator.py:173-264: rotation-bin RANSAC, hypotheses from
    3-point Kabsch on host LAPACK -- the reference's own call, whose null-vector sign decides rotation vs reflection) beside yohoo on the
    kitchen scene (60 clouds, 449 pairs, mutual matcher), each as `n_rep` pipelined passes through run_scenes."""
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    from roreg_amd import synth
    feats, keys, _, pairs = scenes[kitchen]
    out = {'workload': f'kitchen scene ({len(feats)} clouds, {len(pairs)} pairs) x {n_rep} pipelined passes, mutual matcher, max_iter 1000'}
    for ET in ('yohoo', 'yohoc'):
        cfg = default_config(keynum=args.kpts, max_iter=1000, ET=ET)
        et = None
        if ET == 'yohoo':
            et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
        eng = RegistrationEngine(cfg, gf, et); eng.set_gemm_mode(args.gemm)
        job = (feats, keys, pairs, dict(pair_seeds=seeds[kitchen]))
        eng.run_scenes([job] * n_rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.run_scenes([job] * n_rep)[-1]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ok = []
        for r in res:
            gt = synth.pose_transform(scenes[kitchen][2], r.id0, r.id1)
            from roreg_amd.utils.r_eval import compute_R_diff
            ok.append(bool(np.isfinite(r.trans).all() and compute_R_diff(r.trans[:3, :3], gt[:3, :3]) < 15 and np.sqrt(np.sum(np.square(r.trans[:3, 3] - gt[:3, 3]))) < 0.3))
        out[ET] = {'pairs_per_s': len(pairs) * n_rep / dt, 'ms_per_pass': 1e3 * dt / n_rep, 'registration_recall_pointdsc': float(np.mean(ok)),
                   'mean_recalltime': float(np.mean([r.recalltime for r in res]))}
        del eng
    out['yohoc_over_yohoo'] = out['yohoc']['pairs_per_s'] / out['yohoo']['pairs_per_s']
    return out


def rd_rm_nets(cfg):
    """(detector, rotation-coherence matcher, description of their weights): the reference's shipped RD / RM checkpoints when the fixtures hold them."""
    from roreg_amd import synth
    from roreg_amd.network import name2network
    rd = name2network['RD_test'](cfg); rm = name2network['RM_test'](cfg)
    gdir = os.path.join(ROOT, 'tests', 'golden')
    if os.path.exists(f'{gdir}/weights_RD.npz') and os.path.exists(f'{gdir}/weights_RM.npz'):
        rd.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(f'{gdir}/weights_RD.npz').items()})
        rm.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(f'{gdir}/weights_RM.npz').items()})
        return rd, rm, 'the reference\'s shipped RD / RM checkpoints (tests/golden/weights_R{D,M}.npz)'
    synth.seeded_state_dict(rd, 303); synth.seeded_state_dict(rm, 404)
    return rd, rm, 'random-init'


def rd_rm_leg(args, cfg0, gf, et, keynum=2500, n_pairs=100, variants=('fp32', 'bf16', 'fp32_matrix_core_layers')):
    """BASELINE configs[3] / [4] path on one low-overlap scene chunk: detector (RD) -> NMS sampling -> rotation-coherence matcher (RM) at `keynum`
    (2500 = yoho_mat's default; 5000 = `Test.py --keynum 5000`) -> one-shot estimator on the top-`match_n` matches; float32 and bfloat16
    descriptor storage, and the matcher's opt-in matrix-core layers (ROREG_LINEAR_MFMA=1: another rounding, see Match_ot.match_stacked).  Rooflines of the two kernels that dominate
    Match_ot measured with the library's HIP-event brackets: the Sinkhorn iterations (HBM: one pass over every pair's (m+1) x (n+1) float32
    coupling matrix per iteration) and roreg_topk_dot (f32 vector FMA: 2 x 32 x m x n per searched direction)."""
    from roreg_amd import hip, synth
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    n_clouds, overlap = 24, 0.2
    cfg = default_config(keynum=keynum, max_iter=1000, ET='yohoo', RD=True, RM=True)
    rd, rm, weights = rd_rm_nets(cfg)
    feats, keys, poses = synth.make_scene_device(1400, n_clouds, args.kpts, overlap)
    pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(n_clouds, n_pairs, 4343, locality=8.0)]
    seeds = [(11 + zlib.crc32(f'lo:{a}:{b}'.encode())) % (2 ** 32) for a, b in pairs]
    out = {'workload': f'{n_clouds} clouds x {args.kpts} kpts, {n_pairs} pairs, overlap {overlap} (3DLoMatch-like), --RD --RM --ET yohoo --keynum {keynum} --match_n {cfg.match_n}',
           'weights': weights}
    for variant in variants:
        dtype = variant.split('_')[0]
        rm.matrix_core_layers = variant.endswith('matrix_core_layers')
        eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
        eng.set_gemm_mode(args.gemm); eng.set_descriptor_dtype(dtype)
        # the chunk several times through engine.run_scenes, the way the full benchmark runs its scenes: software-pipelined across the scene's host
        # synchronisations (detector scores, NMS neighbour lists, matcher read-outs, match counts, result table), so that the host's rank
        # transforms / NMS selection / task tables of one pass run under the kernels of the next
        job = (feats, keys, pairs, dict(pair_seeds=seeds))
        n_rep = 6
        eng.run_scenes([job] * n_rep)                                    # (warm-up at the timed depth: the caching allocator sees the same live set)
        hip.profile_enable(True); hip.WORK = {}; hip.sinkhorn_iteration_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.run_scenes([job] * n_rep)[-1]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        it_run, it_pairs = hip.sinkhorn_iteration_stats()
        it_frac = it_run / (100.0 * it_pairs) if it_pairs else 1.0           # share of the asked iterations that ran (pairs stop at their float32 fixed point)
        sk_ms, sk_n = hip.profile_read('sinkhorn'); tk_ms, tk_n = hip.profile_read('topk_dot')
        work = hip.WORK; hip.WORK = None; hip.profile_enable(False)
        from roreg_amd.utils.r_eval import compute_R_diff
        ok = []
        for r in res:
            gt = synth.pose_transform(poses, r.id0, r.id1)
            ok.append(bool(np.isfinite(r.trans).all() and compute_R_diff(r.trans[:3, :3], gt[:3, :3]) < 15 and
                           np.sqrt(np.sum(np.square(r.trans[:3, 3] - gt[:3, 3]))) < 0.3))
        leg = {'pairs_per_s': n_pairs * n_rep / dt, 'ms_per_pass': 1e3 * dt / n_rep, 'mean_matches': float(np.mean([r.n_match for r in res])),
               'registration_recall_pointdsc': float(np.mean(ok))}
        if sk_n and work.get('sinkhorn_recompute'):
            # the iterations recompute the scores on the matrix cores (csrc/ot_flash.hip).  One recomputation per element and iteration
            # (of_iter_kernel: seven K = 16 fp16 MFMAs = 224 flop, one v_exp_f32, one add, one fma; a 32-row strip's exponentials stay in registers
            # between the row sums and the column sums) unless ROREG_OT_FUSED=0 or a target cloud exceeds 2559 points (two passes: twice that).
            # No coupling matrix is read; what the fused kernel moves instead is 5 bytes of L2-resident fragments per element and iteration,
            # and that traffic is what bounds it (measured: fetching every fragment twice costs +50-67 %).
            cells = work.get('sinkhorn_cells', 0.0) * it_frac                # (element-iterations that were executed)
            fused = os.environ.get('ROREG_OT_FUSED', '1') != '0' and (cfg.keynum <= 2559 or bool(hip.OT_COOP))
            per_cell = 224.0 if fused else 448.0
            tf = cells * per_cell / (sk_ms * 1e-3) / 1e12
            ex = cells * (1.0 if fused else 2.0) / (sk_ms * 1e-3)
            leg['roofline_sinkhorn'] = {'kernel': ('of_iter_kernel + of_update_kernel (one recomputation per iteration)' if fused else 'of_pass_kernel + of_update_kernel (two passes per iteration)') +
                                                  ': 100 iterations per stacked group of pairs; <s_i, t_j> + potentials + dustbins from seven fp16 MFMAs per 32 x 32 tile, then exponentiated; '
                                                  'the coupling matrix is never read', 'bound': 'l2-fragment-traffic' if fused else 'mfma', 'unit': 'TFLOP/s',
                                        'launch_groups': sk_n, 'avg_ms': sk_ms / sk_n, 'ms_per_pair': sk_ms / max(work.get('sinkhorn_pairs', 1), 1),
                                        'iterations_run_mean': 100.0 * it_frac, 'iterations_asked': 100,
                                        'achieved': tf, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': tf / PEAK_BF16_MFMA_TFLOPS,
                                        'executed_mfma_flop_per_element_and_iteration': per_cell,
                                        'exponentials_per_s': ex, 'exponential_peak_per_s': 256 * 4 * 16 / 2 * 2.4e9,
                                        'exponential_frac': ex / (256 * 4 * 16 / 2 * 2.4e9),
                                        'l2_fragment_bytes_per_s': cells * 5.0 / (sk_ms * 1e-3) if fused else None,
                                        'hbm_bytes_not_read': work.get('sinkhorn_bytes', 0.0),
                                        'note': 'v_exp_f32 issues at half rate on gfx950 (SQ_ACTIVE_INST_VALU, profiles/r04_sinkhorn_fused_pmc.txt); the materialised iteration '
                                                '(ROREG_OT_RECOMPUTE=0) read 4 (m+1)(n+1) bytes per pair and iteration from HBM instead: 0.57 of HBM peak, 0.57-0.59 ms per pair at keynum 2500'}
        elif sk_n:
            gbs = work.get('sinkhorn_bytes', 0.0) / (sk_ms * 1e-3) / 1e9
            leg['roofline_sinkhorn'] = {'kernel': 'ot_fused_pass_kernel + ot_col_merge_kernel (100 iterations per stacked group of pairs; row log-sum-exp and column sums from ONE '
                                                  'read of the coupling matrix per iteration)', 'bound': 'hbm', 'unit': 'GB/s', 'launch_groups': sk_n, 'avg_ms': sk_ms / sk_n,
                                        'ms_per_pair': sk_ms / max(work.get('sinkhorn_pairs', 1), 1),
                                        'achieved': gbs, 'peak': PEAK_HBM_GBS, 'frac': gbs / PEAK_HBM_GBS,
                                        'algorithmic_bytes': 'iterations x sum over pairs of 4 (m+1)(n+1): the kernel\'s one pass per iteration; the reference\'s two '
                                                             'logsumexp passes per iteration (SURVEY 8d: 200 (N+1)^2 4 B) would be twice this figure',
                                        'note': 'a stacked group\'s matrices (32 pairs x 25 MB) exceed the 256 MB Infinity Cache: HBM-bound'}
        if tk_n:
            tf = work.get('topk_flop', 0.0) / (tk_ms * 1e-3) / 1e12
            leg['roofline_topk_dot'] = {'kernel': 'topk_dot_kernel + topk_merge_kernel (k = 16 / 8 / 1 best dot products per row, never materialising the m x n score matrix; '
                                                  'replaces score_mat + full argsort, network/rot_coh_match.py:8-12,34-45)', 'bound': 'valu-f32', 'unit': 'TFLOP/s', 'launches': tk_n,
                                        'avg_ms': tk_ms / tk_n, 'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'frac': tf / PEAK_F32_MFMA_TFLOPS,
                                        'note': '2 x 32 x m x n flop per search on the f32 vector pipe (157.3 TFLOP/s FMA peak); the sorted k-list insertions are extra VALU work'}
        out[variant] = leg
        del eng
    rm.matrix_core_layers = None
    return out


if __name__ == '__main__':
    main()

"""Multi-GPU: one process per GPU; scan pairs shard with no data-path collective; one gather of the result table.

Pairs only connect clouds of the same scene, so the unit of sharding is the scene (each rank extracts only the clouds it
needs and keeps them HBM-resident): scenes are assigned by longest-processing-time bin packing on their pair counts,
and a scene larger than the ideal share is split into contiguous pair ranges.  A cloud of a cut scene that several ranks
need is extracted ONCE, by its owner (the rank holding the first pair of the scene's list that touches it), and its
extractor output `eqv` (38.4 MB) travels to the other ranks point to point over xGMI at the start of the step
(exchange_plan / run_plan; test/extractor.py:47 extracts every cloud once, and so does an N-rank run); the receivers
rebuild the two cheap per-cloud derivatives (matcher descriptor, Des2R coefficients) locally.  The only collective is an
all_gather of the fixed-width float64 result table [pairs, 21] (= backend 'nccl', i.e. RCCL over xGMI, on GPUs; 'gloo'
in CPU tests)."""
import numpy as np
import torch

ROW = 21     # scene, id0, id1, n_match, recalltime, trans[0:15] (row-major 4x4 without the final 1), inlier ratio


def shard_scenes(pair_counts, world_size, cloud_counts=None, cloud_cost=9.0, tolerance=1.02, pair_lists=None, exchange=False, recv_cost=1.0):
    """pair_counts: {scene: n_pairs}; cloud_counts: {scene: n_clouds}.  Extracting a cloud costs about `cloud_cost` pair-units (round 3: a
    least-squares fit of the per-rank times of tools/scaling_estimate.py at N = 1, 2, 4, 8 gives 0.29 ms per pair, 2.65 ms per extracted
    cloud, 0.1-0.4 ms per received cloud: ratios 9 and ~1; rounds 1-2 used 7) and is paid again by every rank that holds a slice of the scene --
    but only for the clouds the slice touches: with pair_lists = {scene: [(id0, id1), ...]} the cost of a range is exact, otherwise
    every slice is charged the whole scene.
    exchange=True (needs pair_lists): a cloud is extracted by ONE rank only -- the one whose range holds the first pair of the scene's
    list that touches it -- and shipped to the other ranges that touch it (exchange_plan); a range [a, b) then pays `cloud_cost` for the
    clouds no earlier pair [0, a) of its scene touches and `recv_cost` (rebuilding the per-cloud derivatives; the transfer itself runs
    beside the rank's whole scenes) for the others.
    -> list (per rank) of [(scene, start, stop)] pair ranges; every pair appears exactly once."""
    cloud_counts = cloud_counts or {s: 0 for s in pair_counts}
    memo = {}

    def clouds_of(p):
        if pair_lists is None or p[0] not in pair_lists:
            return cloud_counts.get(p[0], 0)
        if p not in memo:
            memo[p] = len({int(i) for pr in pair_lists[p[0]][p[1]:p[2]] for i in pr})
        return memo[p]

    seen_before = {}

    def new_clouds_of(p):
        """clouds of range p that no earlier pair of the scene touches (= the ones range p owns under exchange)"""
        if p not in seen_before:
            pl = pair_lists[p[0]]
            first = {}
            for q, pr in enumerate(pl[:p[2]]):
                for i in pr:
                    first.setdefault(int(i), q)
            seen_before[p] = sum(1 for i, q in first.items() if q >= p[1])
        return seen_before[p]

    def cost(p):
        if exchange and pair_lists is not None and p[0] in pair_lists:
            own = new_clouds_of(p)
            return (p[2] - p[1]) + cloud_cost * own + recv_cost * (clouds_of(p) - own)
        return (p[2] - p[1]) + cloud_cost * clouds_of(p)

    pieces = [(s, 0, n) for s, n in pair_counts.items() if n > 0]

    def assign(ps):
        loads = [0.0] * world_size
        out = [[] for _ in range(world_size)]
        for p in sorted(ps, key=lambda q: (-cost(q), q[0], q[1])):
            r = int(np.argmin(loads))
            out[r].append(p)
            loads[r] += cost(p)
        return out, loads

    out, loads = assign(pieces)
    for _ in range(8 * max(world_size, 1)):
        if world_size <= 1 or max(loads) <= tolerance * (sum(loads) / world_size):
            break
        # try cutting every piece in two (at the pair count's midpoint) and pack again; keep the cut that lowers the makespan most.
        # (Cutting only the largest piece of the most loaded rank, as round 1 did, stops at the first cut LPT packs badly.)
        best = None
        for big in pieces:
            if big[2] - big[1] < 2:
                continue
            mid = (big[1] + big[2]) // 2
            cand = [p for p in pieces if p != big] + [(big[0], big[1], mid), (big[0], mid, big[2])]
            c_out, c_loads = assign(cand)
            key = (max(c_loads), sum(c_loads))
            if best is None or key < best[0]:
                best = (key, cand, c_out, c_loads)
        if best is None or best[0][0] >= max(loads) - 1e-9:
            break
        _, pieces, out, loads = best
    if world_size > 1 and max(loads) > tolerance * (sum(loads) / world_size):
        # Wrap-around fill: walk the scenes in a fixed order and fill rank after rank up to a makespan T, cutting the scene that does not
        # fit at the pair where the rank reaches T (the rest goes on to the next rank); the smallest feasible T by bisection.  Every rank
        # then holds whole scenes plus at most two partial ones -- what a packing of whole pieces cannot do when scenes ~ ranks.
        def range_end(sc, a, n, budget):
            """largest b in (a, n] with cost((sc, a, b)) <= budget, or a if even one pair does not fit"""
            lo, hi = a, n
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if cost((sc, a, mid)) <= budget:
                    lo = mid
                else:
                    hi = mid - 1
            return lo

        def fill(order, T):
            plan, r, load = [[] for _ in range(world_size)], 0, 0.0
            for sc in order:
                a, n = 0, pair_counts[sc]
                while a < n:
                    if r >= world_size:
                        return None
                    b = range_end(sc, a, n, T - load)
                    if b == a:                                     # nothing of this scene fits on this rank any more
                        if load == 0.0:
                            return None                            # ... not even on an empty rank: T is too small
                        r += 1; load = 0.0
                        continue
                    plan[r].append((sc, a, b)); load += cost((sc, a, b)); a = b
            return plan

        whole = {sc: cost((sc, 0, n)) for sc, n in pair_counts.items() if n > 0}
        orders = [sorted(whole, key=lambda q: (-whole[q], q)), sorted(whole, key=lambda q: (whole[q], q)), sorted(whole)]
        best_plan, best_T = None, max(loads)
        for order in orders:
            lo, hi = sum(whole.values()) / world_size, max(loads)
            for _ in range(24):
                mid = 0.5 * (lo + hi)
                if fill(order, mid) is not None:
                    hi = mid
                else:
                    lo = mid
            cand = fill(order, hi)
            if cand is not None:
                T = max(sum(cost(p) for p in r) for r in cand)
                if T < best_T - 1e-9:
                    best_plan, best_T = cand, T
        if best_plan is not None:
            out = best_plan
    # contiguous ranges of one scene that landed on the same rank are merged
    for r in range(world_size):
        out[r].sort()
        merged = []
        for p in out[r]:
            if merged and merged[-1][0] == p[0] and merged[-1][2] == p[1]:
                merged[-1] = (p[0], merged[-1][1], p[2])
            else:
                merged.append(p)
        out[r] = merged
    return out


def pack_rows(scene_index, results):
    """[PairResult] -> float64 [n, ROW]."""
    t = np.zeros((len(results), ROW))
    for i, r in enumerate(results):
        t[i, 0] = scene_index; t[i, 1] = float(r.id0); t[i, 2] = float(r.id1); t[i, 3] = r.n_match; t[i, 4] = r.recalltime
        t[i, 5:20] = np.asarray(r.trans, np.float64).reshape(-1)[:15]
        t[i, 20] = getattr(r, 'ir', np.nan)
    return t


def unpack_rows(table):
    out = []
    for row in table:
        T = np.eye(4); T.reshape(-1)[:15] = row[5:20]
        out.append({'scene': int(row[0]), 'id0': str(int(row[1])), 'id1': str(int(row[2])), 'n_match': int(row[3]),
                    'recalltime': int(row[4]), 'trans': T, 'ir': float(row[20])})
    return out


def forced():
    """ROREG_FORCE_COLLECTIVES=1 (bench.py --force-collectives): a one-rank job still issues every collective and sends the extractor outputs of
    a few clouds to itself, so that a single GPU executes the RCCL code path (the all_gather of the table, the grouped send/recv)."""
    import os
    return os.environ.get('ROREG_FORCE_COLLECTIVES', '0') not in ('', '0')


class watchdog:
    """`with watchdog(seconds, 'what'):` -- if the block has not finished after `seconds`, every thread's stack is written to stderr and the
    process ends with status 1 (faulthandler's own C thread: it needs neither the GIL nor a live Python main thread, so it also fires
    while the host is stuck inside a collective or a stream synchronisation behind one).  A hung rank then fails the whole job
    (torch.distributed.run tears the other ranks down) instead of holding the node.  Nothing is re-executed and nothing forks.
    Blocks may nest: leaving the inner one re-arms the outer one with the time it has left.  seconds <= 0 disables.
    The watchdog OWNS faulthandler's one process-global timer while a block is open (another user of dump_traceback_later -- e.g. pytest's
    faulthandler_timeout -- would cancel it and be cancelled by it).  Arming writes one line `what` + the deadline to stderr, so the stacks
    a timeout dumps can be attributed to the collective section that hung."""
    _stack = []

    def __init__(self, seconds, what=''):
        self.seconds, self.what = float(seconds), what

    @staticmethod
    def _arm(deadline, what):
        import faulthandler, os, sys, time
        left = max(deadline - time.monotonic(), 0.05)
        if what and os.environ.get('ROREG_WATCHDOG_QUIET', '0') in ('', '0'):
            sys.stderr.write(f'[roreg watchdog] {what}: stacks + exit(1) if not finished in {left:.0f} s\n')
        sys.stderr.flush()
        faulthandler.dump_traceback_later(left, exit=True)

    def __enter__(self):
        import time
        if self.seconds > 0:
            self.deadline = time.monotonic() + self.seconds
            watchdog._stack.append((self.deadline, self.what))
            self._arm(self.deadline, self.what)
        return self

    def __exit__(self, *exc):
        import faulthandler
        if self.seconds > 0:
            watchdog._stack.pop()
            faulthandler.cancel_dump_traceback_later()
            if watchdog._stack:
                self._arm(*watchdog._stack[-1])
        return False


def collective_timeout(default=60.0):
    import os
    return float(os.environ.get('ROREG_COLLECTIVE_TIMEOUT_S', default))


def init_collectives(backend, rank, world, dev_index=None, timeout_s=None):
    """init_process_group + the group's first world-wide collective (a barrier), under the hang watchdog.  backend 'nccl' = RCCL: the group
    is bound to this rank's GPU (`device_id`), the barrier names it (`device_ids`).  The extractor-output exchange is a batched send/recv in
    which only SOME ranks take part, and a group's first call must involve all of its ranks -- hence the barrier here.
    Works at world size 1 too (forced collectives on a single GPU)."""
    import datetime, os, socket
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in os.environ:
        if world != 1:
            raise RuntimeError('init_collectives: MASTER_PORT is not set (start the ranks with torch.distributed.run)')
        s = socket.socket(); s.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s.getsockname()[1]); s.close()
    # generous: the ranks of a fresh box import torch at different speeds (minutes apart in the worst case) before they meet here
    t = collective_timeout(600.0) if timeout_s is None else timeout_s
    with watchdog(t, 'init_process_group + first barrier'):
        kw = {'device_id': torch.device('cuda', dev_index)} if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(t, 30.0) * 4), **kw)
        dist.barrier(device_ids=[dev_index]) if backend == 'nccl' else dist.barrier()
    return dist


def gather_table(local, device=None, counts=None, force=None, group=None):
    """all_gather of ragged [n_r, ROW] float64 tables -> [sum n_r, ROW] on every rank (rank order).
    counts: the number of rows of every rank when it is known beforehand (it is: every rank derives the same shard plan) -- then this is
    ONE collective; without it the sizes are exchanged first.
    A one-rank group returns `local` without a collective unless force (default: forced()) -- then the same calls run with one rank."""
    import torch.distributed as dist
    force = forced() if force is None else force
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return np.asarray(local, np.float64).reshape(-1, ROW)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else ('cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
    if counts is None:
        n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        counts = [int(c.item()) for c in sizes]
    elif len(local) != counts[rank]:
        raise ValueError(f'gather_table: rank {rank} holds {len(local)} rows, the plan says {counts[rank]}')
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, ROW), dtype=torch.float64, device=dev)
    if len(local):
        buf[:len(local)] = torch.as_tensor(np.asarray(local, np.float64), device=dev)
    out = torch.empty((world * mx, ROW), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)            # (one call path for nccl and gloo: the CPU tests exercise it)
    out = out.view(world, mx, ROW).cpu().numpy()
    return np.concatenate([out[r, :c] for r, c in enumerate(counts)], 0).reshape(-1, ROW)


# ---- cut scenes: every cloud is extracted once, its `eqv` travels point to point ------------------------------------------------------
def exchange_plan(plan, pair_lists):
    """plan: shard_scenes() output; pair_lists {scene: [(id0, id1)]}.
    -> (owner {(scene, cloud): rank} for the clouds of scenes held by more than one rank, transfers [(scene, cloud, src, dst)] in one
    canonical (sorted) order that every rank derives identically).  The owner of a cloud is the rank whose pair range holds the first
    pair of the scene's list that touches it -- the rule shard_scenes(exchange=True) prices."""
    where = {}
    for r, pieces in enumerate(plan):
        for s, a, b in pieces:
            where.setdefault(s, []).append((a, b, r))
    owner, transfers = {}, []
    for s in sorted(where):
        ranges = sorted(where[s])
        if len({r for _, _, r in ranges}) == 1:
            continue                                              # the whole scene lives on one rank
        need = {}
        for a, b, r in ranges:
            for pr in pair_lists[s][a:b]:
                for i in pr:
                    i = int(i)
                    owner.setdefault((s, i), r)
                    need.setdefault(i, set()).add(r)
        for i in sorted(need):
            transfers += [(s, i, owner[(s, i)], d) for d in sorted(need[i] - {owner[(s, i)]})]
    return owner, transfers


def extractions_per_rank(plan, pair_lists, exchange=True):
    """Cloud extractions every rank performs in one pass of `plan` (diagnostics / tests): with the exchange a cloud is extracted by its
    owner only, without it by every rank whose ranges touch it."""
    owner, _ = exchange_plan(plan, pair_lists) if exchange else ({}, [])
    out = []
    for r, pieces in enumerate(plan):
        touched = {(s, int(i)) for s, a, b in pieces for pr in pair_lists[s][a:b] for i in pr}
        out.append(sum(1 for c in touched if owner.get(c, r) == r))
    return out


class EqvExchange:
    """The point-to-point exchange of extractor outputs at the start of a step: one torch.distributed.batch_isend_irecv (backend nccl = RCCL:
    one grouped launch, device buffers, xGMI peer-to-peer; the sends read `eqv` on the stream it was produced on, the transfer itself runs on
    the communicator's stream beside this rank's whole scenes).  Under gloo (CPU tests; the shared-GPU control-flow check) the payload is
    staged through host memory.  A transfer whose source and destination are the same rank (forced collectives on one GPU) is a send AND a
    receive of that rank inside the one group."""

    def __init__(self, rank, device_payloads=None, group=None):
        import torch.distributed as dist
        self.rank = rank
        self.group = group
        self.on_device = (dist.get_backend(group) == 'nccl') if device_payloads is None else device_payloads
        self.works, self.landed = [], []
        self.bytes_sent = self.bytes_received = 0

    def start(self, transfers, get_eqv, alloc):
        """transfers: the canonical list (all ranks'); get_eqv(scene, cloud) -> the tensor to send; alloc(scene, cloud) -> the tensor to
        receive into (device tensors; staged here when the backend cannot move them)."""
        import torch.distributed as dist
        ops, keep = [], []
        for tag, (s, i, src, dst) in enumerate(transfers):
            if src == dst == self.rank and not self.on_device:       # (gloo has no pair to itself: a host-staged self-transfer is the staging copy alone)
                t = get_eqv(s, i); buf = alloc(s, i)
                self.landed.append(((s, i), buf, t.cpu()))
                self.bytes_sent += t.numel() * t.element_size(); self.bytes_received += buf.numel() * buf.element_size()
                continue
            if src == self.rank:
                t = get_eqv(s, i)
                t = t if self.on_device else t.cpu()
                keep.append(t)
                self.bytes_sent += t.numel() * t.element_size()
                ops.append(dist.P2POp(dist.isend, t, dst, group=self.group, tag=tag))
            if dst == self.rank:
                buf = alloc(s, i)
                stage = buf if self.on_device else torch.empty(buf.shape, dtype=buf.dtype)
                self.landed.append(((s, i), buf, stage))
                self.bytes_received += buf.numel() * buf.element_size()
                ops.append(dist.P2POp(dist.irecv, stage, src, group=self.group, tag=tag))
        self.keep = keep
        self.works = dist.batch_isend_irecv(ops) if ops else []

    def wait(self):
        """Orders the current stream behind the transfers (nccl) / blocks until they have arrived (gloo) -> {(scene, cloud): eqv tensor}."""
        for w in self.works:
            w.wait()
        out = {}
        for key, buf, stage in self.landed:
            if stage is not buf:
                buf.copy_(stage)
            out[key] = buf
        self.works, self.landed, self.keep = [], [], []
        return out


def self_transfers(pieces, pair_lists, rank=0, n_clouds=4):
    """Forced collectives on one rank: the first `n_clouds` clouds of the LAST scene of `pieces` are 'shipped' from this rank to itself
    (extract -> isend -> irecv -> rebuild the derivatives), which drives run_plan's whole exchange branch with a one-rank group."""
    if not pieces:
        return []
    s, a, b = pieces[-1]
    ids = []
    for pr in pair_lists[s][a:b]:
        for i in pr:
            if int(i) not in ids:
                ids.append(int(i))
    return [(s, i, rank, rank) for i in sorted(ids[:n_clouds])]


def run_plan(engine, pieces, scene_inputs, transfers=(), rank=0, exchange=None, min_jobs=None, min_pairs=48, stats=None, seeded=None, **run_kw):
    """One pass of this rank's share of a shard plan.  pieces [(scene, a, b)]; scene_inputs(scene) -> (feats, keys, pair_ids, pair_seeds or
    None) with feats / keys indexable by int cloud id; transfers: exchange_plan()'s list (empty: every rank extracts what it touches).
    Order: (0) the clouds this rank owns and others need are extracted and sent, the receives are posted; (1) the scenes this rank
    holds without imports; (2) the pair ranges that wait for imported clouds.  -> [(scene, a, b, [PairResult])] in `pieces` order.
    stats (a dict, optional) receives 'eqv_bytes_sent' / 'eqv_bytes_received' of this pass.
    seeded: whether scene_inputs() hands out per-pair seeds (the condition for software-pipelining the jobs); None = ask scene_inputs for
    every piece up front (fine when that is cheap; a driver whose scene_inputs reads files passes the flag instead, so that a scene's
    inputs are touched only when the pipeline reaches it)."""
    if min_jobs is None:
        import os
        min_jobs = int(os.environ.get('ROREG_PLAN_MIN_JOBS', 2))   # (the switch is for A/B measurements: 1 / 2 / 4 measured alike, profiles/r03_split_ab.txt)
    sends = [t for t in transfers if t[2] == rank]
    recvs = [t for t in transfers if t[3] == rank]
    cache = {}                                                     # scene -> {cloud: CloudState}: exported, imported, reused across ranges
    shared = {s for s, _, _, _ in transfers}
    ex = None
    if sends or recvs:
        by_scene = {}
        for s, i, _, _ in sends:
            by_scene.setdefault(s, set()).add(i)
        for s in sorted(by_scene):
            feats, keys = scene_inputs(s)[:2]
            ids = sorted(by_scene[s])
            cache.setdefault(s, {}).update(zip(ids, engine.extract_many([feats[i] for i in ids], [keys[i] for i in ids])))
        ex = exchange if exchange is not None else EqvExchange(rank)
        ex.start(transfers, lambda s, i: cache[s][i].eqv,
                 lambda s, i: engine.alloc_eqv(scene_inputs(s)[0][i]))
    importing = {s for s, _, _, _ in recvs}
    order = sorted(range(len(pieces)), key=lambda q: (pieces[q][0] in importing, q))
    out = [None] * len(pieces)

    # One software pipeline over all of this rank's jobs (engine.run_scenes), the ranges that use received clouds last: their jobs are
    # created lazily, so the wait for the transfers (a stream-ordered wait under nccl) is enqueued only when the pipeline reaches them --
    # behind the whole scenes' kernels -- and the pipeline does not drain in between.  A rank that holds only one or two scenes (8 ranks on
    # 8 scenes) has nothing to hide a scene's two host synchronisations behind, so its pair ranges are halved (down to `min_pairs`) until
    # `min_jobs` jobs are in flight (default 2: a lone scene becomes two jobs): the halves of a scene share one cloud cache -- the second finds the clouds the first extracted (the
    # cache is filled when the first half's extraction is ENQUEUED, and the stream orders the kernels) -- and every pair's result is
    # independent of the batch it is computed in, so the results are bitwise those of the unsplit range.
    jobs = []                                                      # [piece index, a, b, cloud cache]
    for q in order:
        s, a, b = pieces[q]
        jobs.append([q, a, b, cache.setdefault(s, {}) if s in shared else None])
    if seeded is None:
        seeded = all(scene_inputs(pieces[q][0])[3] is not None for q in order)
    pipelined = hasattr(engine, 'run_scenes') and bool(seeded)
    if pipelined:
        while len(jobs) < min_jobs:
            k = max(range(len(jobs)), key=lambda i: jobs[i][2] - jobs[i][1])
            q, a, b, rd = jobs[k]
            if b - a < 2 * min_pairs:
                break
            rd = {} if rd is None else rd
            mid = (a + b) // 2
            jobs[k:k + 1] = [[q, a, mid, rd], [q, mid, b, rd]]
    state = {'waited': not recvs}

    def call_of(job):
        q, a, b, rd = job
        s = pieces[q][0]
        if s in importing and not state['waited']:
            for (sc, i), eqv in ex.wait().items():
                f, k = scene_inputs(sc)[:2]
                cache.setdefault(sc, {})[i] = engine.cloud_from_eqv(f[i], eqv, k[i])
            state['waited'] = True
        feats, keys, pairs, seeds = scene_inputs(s)
        return feats, keys, pairs[a:b], dict(pair_seeds=None if seeds is None else seeds[a:b], ready=rd, **run_kw)

    if pipelined:
        res = engine.run_scenes([(lambda job=job: call_of(job)) for job in jobs])
    else:
        res = []
        for job in jobs:
            f, k, p, kw = call_of(job)
            res.append(engine.run_scene(f, k, p, **kw))
    for q in order:
        out[q] = (pieces[q][0], pieces[q][1], pieces[q][2], [r for (qq, _, _, _), rs in zip(jobs, res) if qq == q for r in rs])
    waited = state['waited']
    if ex is not None and not waited:                              # a rank that only sends: its sends complete before the step ends
        ex.wait()
    if stats is not None:
        stats['eqv_bytes_sent'] = stats.get('eqv_bytes_sent', 0) + (ex.bytes_sent if ex is not None else 0)
        stats['eqv_bytes_received'] = stats.get('eqv_bytes_received', 0) + (ex.bytes_received if ex is not None else 0)
    return out

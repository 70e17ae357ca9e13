"""Device-resident scene pipeline: extractor -> [detector] -> matcher -> estimator with every
per-cloud tensor kept in HBM for the whole scene, so stage boundaries are views instead of .npy round trips
(the reference re-loads 8 x 38.4 MB per pair).

Stage order and host-RNG consumption are the reference's (test/evaluator.py:39-48): all clouds are extracted,
then ALL pairs are matched (np.random.shuffle x2 per pair when the detector is off, matcher.py:83-88), then
ALL pairs are estimated (one np.random.shuffle of the hypothesis order per pair, estimator.py:423-425).  The
only device->host syncs are one per scene for the match counts (the hypothesis shuffle needs M) and one for
the result table.
"""
from dataclasses import dataclass, field

import os
import time

import numpy as np
import torch

from . import hip


@dataclass
class CloudState:
    before: torch.Tensor            # FCGF-like input group features [N,32,60] f32 (device)
    eqv: torch.Tensor = None        # GF output [N,32,60]
    eqv_ft: torch.Tensor = None     # group-Fourier coefficients of eqv [N,32,60] f32 (hip.feat_coefs): operand of the irrep-domain Des2R
    inv: torch.Tensor = None        # matcher invariant descriptor [N,32]
    keys: torch.Tensor = None       # keypoints [N,3] f64 (device)
    det: np.ndarray = None          # detector rank scores (host, as det_score/*.npy)
    keys_host: np.ndarray = None    # keypoints on the host (downloaded on first use: the yohoc estimator's 3-point Kabsch runs on LAPACK)
    nms: dict = field(default_factory=dict)   # keynum -> NMS sample of this cloud (a pure function of the cloud)


_POOL = None


def _host_pool():
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max(1, min(64, (os.cpu_count() or 2) // 2)))     # LAPACK releases the GIL: the pairs' 3-point Kabsch stacks run side by side
    return _POOL


@dataclass
class PairResult:
    id0: str
    id1: str
    n_match: int
    trans: np.ndarray               # [4,4] f64
    recalltime: int
    matches: torch.Tensor = None    # [M,2] int64 (device)
    scores: np.ndarray = None


class StageFileWriter:
    """Asynchronous writer of the reference's inter-stage files for the device-resident engine:

        {cache}/{scene}/match_{keynum}/{a}-{b}.npy            [M,2] int64     test/matcher.py:108, :209
        {cache}/{scene}/match_{keynum}/scores/{a}-{b}.npy     [M] f64 ones (mutual) / f32 (rotation-coherence matcher)   matcher.py:109, :210
        {cache}/{scene}/match_{keynum}/DR_index/{a}-{b}.npy   [M] int64       test/estimator.py:111
        {cache}/{scene}/match_{keynum}/Trans_pre/{a}-{b}.npy  [M,3,4] f64     test/estimator.py:367

    byte for byte what the file-coupled stage classes write.  Off the critical path: a device tensor is copied to pinned host memory on
    a side stream (ordered after its producing kernels by an event), and a worker thread waits for that copy and calls np.save; the
    engine never synchronises for it.  close() joins the worker (call it before reading the files).

    clouds_dir (optional; `{cache}/{feature scene}`): the per-cloud stage files are written too --

        {clouds_dir}/YOHO_Output_Group_feature/{pc}.npy       [N,32,60] f32   test/extractor.py:57  (every cloud the engine extracts)
        {clouds_dir}/det_score/{pc}.npy                       [N] f32         test/detector.py:45-47 (every cloud it scores)

    which makes the engine a complete stand-in for the stage chain of test/evaluator.py:39-48 (roreg_amd/test/evaluator.py)."""

    def __init__(self, cfg, dataset_name, keynum, clouds_dir=None):
        import queue
        import threading
        from .utils.utils import make_non_exists_dir
        self.dir = f'{cfg.output_cache_fn}/{dataset_name}/match_{keynum}'
        for d in ('', '/scores', '/DR_index', '/Trans_pre'):
            make_non_exists_dir(self.dir + d)
        self.clouds_dir = clouds_dir
        if clouds_dir is not None:
            make_non_exists_dir(f'{clouds_dir}/YOHO_Output_Group_feature')
            if getattr(cfg, 'RD', False):
                make_non_exists_dir(f'{clouds_dir}/det_score')
        # high priority: the copy kernels of the 38.4 MB feature downloads otherwise queue for compute units behind the estimator's launches
        self.stream = hip.named_stream('stage_file_writer', priority=-1)
        self.q = queue.Queue()
        self.error = None
        self.log = []                                                # (directory, bytes, time the copy had landed, time the file was written)
        # a few workers: np.save of a contiguous array is one fwrite with the GIL released, so the 38.4 MB feature files of a scene go to
        # the page cache side by side (one worker wrote 60 of them in ~2 s, longer than the scene's kernels take)
        self.threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, int(os.environ.get('ROREG_WRITER_THREADS', 4))))]
        for t in self.threads:
            t.start()
        self.also_precreate = []                                     # (a caller's own per-pair files, e.g. the evaluator's result archives: created after the engine's)
        self._creator, self._cq = None, queue.Queue()

    def precreate(self, paths):
        """Create (empty) files that are ALWAYS rewritten -- match lists, scores, DR_index, Trans_pre, result archives; never the extractor /
        detector outputs, whose existence means 'done' to the stage chain -- ahead of their data, on ONE thread of its own.  Creating a file
        on the GPU box's overlay filesystem is a serialised ~0.16 ms (0.33 when several threads do it at once) against 0.01-0.03 ms for
        writing into one that exists (tools/probe/small_files_probe.py): a 449-pair scene's ~2250 creations otherwise queue up behind the
        scene's last kernel (0.13-0.35 s).  A file the writer got to first is left as it is (no truncation here)."""
        import threading
        if os.environ.get('ROREG_WRITER_PRECREATE', '1') == '0':     # (the switch is for A/B measurements)
            return
        if self._creator is None:
            def create():
                while True:
                    batch = self._cq.get()
                    if batch is None:
                        return
                    for p in batch:
                        try:
                            os.close(os.open(p, os.O_CREAT | os.O_WRONLY, 0o644))
                        except OSError:
                            pass                                     # (the write that follows reports what is wrong with the path)
            self._creator = threading.Thread(target=create, daemon=True)
            self._creator.start()
        self._cq.put(list(paths))

    def precreate_pairs(self, pair_ids, trans_pre):
        """the engine's per-pair files of a scene, in the order they will be written, then the caller's (also_precreate)"""
        names = [f'{a}-{b}.npy' for a, b in pair_ids]
        self.precreate([f'{self.dir}/{sub}{n}' for sub in ('', 'scores/', 'DR_index/') + (('Trans_pre/',) if trans_pre else ()) for n in names] + list(self.also_precreate))
        self.also_precreate = []

    def _work(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                if callable(item):                                   # submit(): any host-side file work
                    item()
                    continue
                if len(item) == 5:                                   # save_many: one pinned buffer, a file per slice
                    paths, host, done, shapes, buf = item
                    done.synchronize()
                    t_landed = time.perf_counter()
                    a, o, parts = host.numpy(), 0, []
                    for shape in shapes:
                        n = int(np.prod(shape))
                        parts.append(a[o:o + n].reshape(shape)); o += n
                    hip.write_npy_files(paths, parts, n_threads=2)   # (np.save's bytes without np.save's ~80 us of interpreter time per file)
                    self.log.append((os.path.basename(os.path.dirname(paths[0])), a.nbytes, t_landed, time.perf_counter()))
                    del a, parts, host
                    hip.pinned_pool.release(buf)                     # (the copy has landed and the files are written)
                    continue
                path, host, done, buf = item
                if done is not None:
                    done.synchronize()
                t_landed = time.perf_counter()
                a = host.numpy() if torch.is_tensor(host) else host
                hip.write_npy_files([path], [a], n_threads=1)
                self.log.append((os.path.basename(os.path.dirname(path)), a.nbytes, t_landed, time.perf_counter()))
                del a, host
                if buf is not None:
                    hip.pinned_pool.release(buf)
            except Exception as e:                                  # surfaced by close()
                self.error = e

    def save(self, rel, a):
        """rel: path below match_{keynum}/ without '.npy'; a: device tensor or host ndarray (dtype as it must appear on disk)."""
        self.save_path(f'{self.dir}/{rel}.npy', a)

    def save_path(self, path, a):
        if not torch.is_tensor(a):
            self.q.put((path, np.ascontiguousarray(a), None, None))
            return
        ready = torch.cuda.Event(); ready.record()                  # after the producing kernels on the current stream
        host, buf = self._pinned(a)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            host.copy_(a, non_blocking=True)
            done = torch.cuda.Event(); done.record(self.stream)
        a.record_stream(self.stream)                                # the allocator must not hand the block out before the copy ran
        self.q.put((path, host, done, buf))

    @staticmethod
    def _pinned(a):
        """a pinned host tensor of a's shape and dtype out of the process's pool (hip.PinnedPool: no pinned allocation in the steady state)"""
        nbytes = a.numel() * a.element_size()
        buf = hip.pinned_pool.acquire(nbytes)
        return buf[:nbytes].view(a.dtype).reshape(a.shape), buf

    def submit(self, fn):
        """run fn() on a writer thread (result files, logs); close() waits for it"""
        self.q.put(fn)

    def save_many_host(self, rels, arrays):
        """Many host arrays (a scene's per-pair score files), one queue entry, one threaded C call."""
        if not len(rels):
            return
        paths = [f'{self.dir}/{r}.npy' for r in rels]

        def write():
            t0 = time.perf_counter()
            hip.write_npy_files(paths, arrays, n_threads=2)
            self.log.append((os.path.basename(os.path.dirname(paths[0])), sum(a.nbytes for a in arrays), t0, time.perf_counter()))
        self.q.put(write)

    def save_many(self, rels, tensors):
        """Many small device tensors of one dtype (a scene's per-pair match lists, DR_index, Trans_pre): ONE concatenation, ONE device -> pinned-host
        copy and one queue entry; the worker writes each file from its slice (a copy per file cost ~0.3 ms of host time on the critical path:
        ~1800 of them per 449-pair scene)."""
        if not tensors:
            return
        shapes = [tuple(t.shape) for t in tensors]
        flat = torch.cat([t.reshape(-1) for t in tensors])
        ready = torch.cuda.Event(); ready.record()
        host, buf = self._pinned(flat)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            host.copy_(flat, non_blocking=True)
            done = torch.cuda.Event(); done.record(self.stream)
        flat.record_stream(self.stream)
        self.q.put(([f'{self.dir}/{r}.npy' for r in rels], host, done, shapes, buf))

    def close(self):
        if self._creator is not None:
            self._cq.put(None)
            self._creator.join()
            self._creator = None
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.error is not None:
            raise self.error


class _LazySeq:
    """[container[i] for i in ids] evaluated item by item, when asked (extract_many fetches a batch's clouds right before their launch)."""

    def __init__(self, container, ids):
        self.container, self.ids = container, ids

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, q):
        return self.container[self.ids[q]]


def even_groups(sizes, cap):
    """Consecutive groups [(i, j)] of `sizes` whose sums stay <= cap (a single item above the cap is a group of its own), EVEN in size:
    100 pairs of 2500 points under a cap of 80,000 are 4 x 25, not 32 + 32 + 32 + 4 -- a small last group leaves most of the chip idle, and
    a pair's results do not depend on what is stacked beside it."""
    sizes = [int(x) for x in sizes]
    if not sizes:
        return []
    n_groups = max(1, -(-sum(sizes) // int(cap)))
    target = sum(sizes) / n_groups
    out, i = [], 0
    while i < len(sizes):
        j, pts = i, 0
        while j < len(sizes) and (j == i or (pts + sizes[j] <= cap and pts + sizes[j] / 2 <= target)):
            pts += sizes[j]; j += 1
        out.append((i, j)); i = j
    return out


class RegistrationEngine:
    def __init__(self, cfg, gf_net, et_net, rd_net=None, rm_net=None):
        self.cfg = cfg
        self.gf = gf_net
        self.et = et_net            # ET_test (None is fine for the yohoc estimator, which never evaluates it)
        self.rd = rd_net            # detector_eqv_test (needed when cfg.RD)
        self.rm = rm_net            # Match_ot (needed when cfg.RM)
        hip.ensure_tables()
        # points per side stacked into one pass of the rotation-coherence matcher: 104 pairs at keynum 2500 (~6 GB of Sinkhorn read-out matrices and
        # activations).  Measured on BASELINE configs[3]'s chunk (24 clouds / 100 pairs, one box): 80,000 -> 572-591 pairs/s, 160,000 -> 609-616,
        # 260,000 -> 629-633: every kernel of the matcher fills the chip better, the Sinkhorn iterations drop from 0.37 to 0.32 ms per pair
        self.rm_max_points = int(os.environ.get('ROREG_RM_MAX_POINTS', 260000))
        self.feat_dtype = torch.bfloat16 if getattr(cfg, 'dtype', 'fp32') == 'bf16' else torch.float32
        self.extract_rows = int(os.environ.get('ROREG_EXTRACT_ROWS', 65536))    # keypoints per extractor launch (activations: ~370 KB per keypoint at peak)
        self.lt_rows = int(os.environ.get('ROREG_LT_ROWS', 131072))                  # correspondences per pass of the ET network (local_transforms_many)
        self.phase_ms = None        # set to {} to collect synchronised wall times per phase of run_scene (diagnostics only)
        self._side = None           # side stream of run_scenes' downloads

    def _mark(self, name, t0):
        """Diagnostics: with phase_ms set, synchronise and add the wall time since t0 to phase `name`; returns the new t0."""
        if self.phase_ms is None:
            return t0
        import time
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.phase_ms[name] = self.phase_ms.get(name, 0.0) + 1e3 * (t1 - t0)
        return t1

    def set_descriptor_dtype(self, name):
        """'fp32' | 'bf16' (BASELINE config 5): device storage of the group features -- the FCGF-like input is rounded to bfloat16 once when it
        is taken in, the extractor's last kernel stores its output in bfloat16, and every consumer (matcher descriptor, Des2R, ET input
        assembly, detector, rotation-coherence matcher) reads the stored values and accumulates in float32.  Parity is then defined on
        the bf16-rounded tensors: indices equal the oracle's fed those tensors (tests/test_hip_bf16.py)."""
        if name not in ('fp32', 'bf16'):
            raise ValueError("descriptor dtype must be 'fp32' or 'bf16'")
        self.feat_dtype = torch.bfloat16 if name == 'bf16' else torch.float32

    def set_gemm_mode(self, mode):
        """'f16x2' | 'bf16x3' | 'f32': how the group-conv GEMMs / convolutions feed the matrix cores (hip.GEMM_MODE)."""
        from .network.gf_fourier import FourierGF
        if mode == 'split':
            mode = 'bf16x3'
        if mode not in hip.GEMM_MODES:
            raise ValueError(f'gemm mode must be one of {hip.GEMM_MODES}')
        net = self.gf.PartI_net
        if net._fourier is None:
            object.__setattr__(net, '_fourier', FourierGF(net))
        net._fourier.gemm = mode
        if self.et is not None:
            self.et.gemm = mode
        if self.rd is not None:
            if self.rd._fourier is None:
                from .network.gf_fourier import FourierRD
                object.__setattr__(self.rd, '_fourier', FourierRD(self.rd.eqv_encoder[0]))
            self.rd._fourier.gemm = mode

    # ---- per cloud ---------------------------------------------------------------------------------------
    def extract(self, feats, keys):
        """feats: [N,32,60] f32 (host ndarray or device tensor); keys [N,3] f64."""
        x = feats if torch.is_tensor(feats) else torch.from_numpy(np.ascontiguousarray(feats, np.float32))
        x = x.to('cuda', torch.float32).to(self.feat_dtype).contiguous()
        with torch.no_grad():
            eqv = self.gf.PartI_net(x, want_inv=False, out_dtype=self.feat_dtype)['eqv']
        k = keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys, np.float64))
        return CloudState(before=x, eqv=eqv, eqv_ft=hip.feat_coefs(eqv), inv=hip.inv_descriptor(eqv), keys=k.to('cuda', torch.float64).contiguous())

    def extract_many(self, feats_list, keys_list, max_rows=None, on_batch=None):
        """Several clouds per group-conv launch: a 5000-keypoint cloud is 9.2 waves of workgroups on the 512 resident slots,
        so a lone cloud wastes ~8 % in the partial last wave; batching clouds makes that tail negligible."""
        max_rows = self.extract_rows if max_rows is None else max_rows
        def on_device(f):                                        # (tensors that already sit on the device in the storage type pass untouched:
            if torch.is_tensor(f) and f.is_cuda and f.dtype == self.feat_dtype:      #  a no-op `.to()` still costs ~60 us of host time each)
                return f
            if torch.is_tensor(f) and not f.is_cuda and f.is_pinned():               # (a pinned host tensor -- the evaluator's prefetching loader -- goes up asynchronously)
                return f.to('cuda', non_blocking=True).to(torch.float32).to(self.feat_dtype)
            return (f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f, np.float32))).to('cuda', torch.float32).to(self.feat_dtype)
        out = []
        i = 0
        n_all = len(keys_list)
        fetched = {}                                             # (a lazy mapping -- files behind it -- is asked for each cloud once)
        def get(q):
            if q not in fetched:
                fetched[q] = feats_list[q]
            return fetched[q]
        while i < n_all:
            j, rows = i, 0
            while j < n_all and (j == i or rows + int(get(j).shape[0]) <= max_rows):
                rows += int(get(j).shape[0]); j += 1
            xs = [on_device(fetched.pop(q)) for q in range(i, j)]      # uploaded batch by batch: the next batch's host reads run under this batch's kernels
            # clouds that already sit back to back in one allocation (bench.py's device-generated scenes, a preloaded scene buffer) are
            # used in place; otherwise one concatenating copy
            adjacent = all(xs[q].is_contiguous() and xs[q].dtype == xs[0].dtype and xs[q].untyped_storage().data_ptr() == xs[0].untyped_storage().data_ptr() and
                           xs[q].data_ptr() == xs[q - 1].data_ptr() + xs[q - 1].numel() * xs[q - 1].element_size() for q in range(1, j - i)) and xs[0].is_contiguous()
            if j - i == 1:
                xcat = xs[0].contiguous()
            elif adjacent:
                xcat = torch.as_strided(xs[0], (rows, 32, 60), (1920, 60, 1))
            else:
                xcat = torch.cat(xs, 0)
            with torch.no_grad():
                eqv = self.gf.PartI_net(xcat, want_inv=False, out_dtype=self.feat_dtype)['eqv']
            inv = hip.inv_descriptor(eqv)
            eft = hip.feat_coefs(eqv)
            # host keypoints (the evaluator's files): ONE upload per batch through the staging ring -- a pageable .to('cuda') per cloud waits for
            # every kernel queued so far (the host then never runs ahead of the extractor), and a ring slot per cloud wraps the ring inside a scene
            on_host = [q for q in range(i, j) if not (torch.is_tensor(keys_list[q]) and keys_list[q].is_cuda)]
            k_dev = {}
            if on_host:
                hk = [np.ascontiguousarray(keys_list[q].numpy() if torch.is_tensor(keys_list[q]) else keys_list[q], np.float64).reshape(-1, 3) for q in on_host]
                flat_k, ko = hip.upload(np.concatenate(hk)), 0
                for q, h in zip(on_host, hk):
                    k_dev[q] = flat_k[ko:ko + h.shape[0]]; ko += h.shape[0]
            o = 0
            for q in range(i, j):
                n = xs[q - i].shape[0]
                k = k_dev[q] if q in k_dev else keys_list[q]
                if not (k.dtype == torch.float64 and k.is_contiguous()):
                    k = k.to(torch.float64).contiguous()
                out.append(CloudState(before=xcat[o:o + n], eqv=eqv[o:o + n], eqv_ft=eft[o:o + n], inv=inv[o:o + n], keys=k))
                o += n
            if on_batch is not None:                             # (the writer: a batch's downloads start behind ITS kernels, not behind the whole scene's)
                on_batch(range(i, j), out[i:j])
            i = j
        return out

    def alloc_eqv(self, before):
        """Receive buffer for the extractor output of the cloud whose input features are `before` ([N,32,60])."""
        return torch.empty((int(before.shape[0]), 32, 60), dtype=self.feat_dtype, device='cuda')

    def cloud_from_eqv(self, before, eqv, keys):
        """CloudState of a cloud whose extractor output `eqv` was computed elsewhere (multi-GPU: the owner rank of a cut scene's cloud
        ships it over xGMI, distributed.run_plan): only the two cheap per-cloud derivatives are rebuilt here.  Bitwise the state
        extract_many() would have produced (the extractor's block scales are per keypoint, so `eqv` does not depend on who computed it)."""
        x = before if torch.is_tensor(before) else torch.from_numpy(np.ascontiguousarray(before, np.float32))
        if not (x.is_cuda and x.dtype == self.feat_dtype):
            x = x.to('cuda', torch.float32).to(self.feat_dtype)
        k = keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys, np.float64))
        if not (k.is_cuda and k.dtype == torch.float64 and k.is_contiguous()):
            k = k.to('cuda', torch.float64).contiguous()
        eqv = eqv.to('cuda', self.feat_dtype).contiguous()
        return CloudState(before=x.contiguous(), eqv=eqv, eqv_ft=hip.feat_coefs(eqv), inv=hip.inv_descriptor(eqv), keys=k)

    def detect(self, cloud):
        """raw std scores -> rank/N on the host exactly as test/detector.py:45-46."""
        self.detect_many([cloud])
        return cloud.det

    @staticmethod
    def _drive(gen):
        """Run a stage generator synchronously: every tensor list it yields is downloaded with a blocking copy and sent back as numpy arrays."""
        try:
            req = next(gen)
            while True:
                req = gen.send([t.cpu().numpy() for t in req])
        except StopIteration as fin:
            return fin.value

    def detect_many(self, clouds):
        """Detector scores of several clouds: every cloud's network pass is enqueued, ONE download, then the rank transform of
        test/detector.py:45-46 per cloud on the host."""
        return self._drive(self._detect_steps(clouds))

    def _detect_steps(self, clouds):
        """detect_many as a generator (yields the device tensor it needs on the host, is sent its numpy copy): inside run_scenes the download
        is asynchronous and the rank transforms run under the kernels of the other scenes in flight."""
        todo = [c for c in clouds if c.det is None]
        if not todo:
            return
        raw, i = [], 0
        with torch.no_grad():
            while i < len(todo):                                      # several clouds per pass of the (per-keypoint) detector network
                j, rows = i, 0
                while j < len(todo) and (j == i or rows + todo[j].eqv.shape[0] <= 65536):
                    rows += todo[j].eqv.shape[0]; j += 1
                x = todo[i].eqv if j - i == 1 else torch.cat([c.eqv for c in todo[i:j]])
                raw.append(self.rd({'feats': x})['scores'])
                i = j
        flat = (yield [torch.cat(raw)])[0]
        offs = np.cumsum([0] + [c.eqv.shape[0] for c in todo])

        def rank(q):                                                  # (numpy's sort releases the GIL: the clouds' rank transforms run side by side)
            s = flat[offs[q]:offs[q + 1]].copy()
            a = np.argsort(s)
            s[a] = np.arange(s.shape[0]) / s.shape[0]
            return s
        for c, s in zip(todo, _host_pool().map(rank, range(len(todo)))):
            c.det = s

    def nms_many(self, clouds, keynum):
        """NMS sampling (test/matcher.py:11-42) of several clouds: it is a pure function of the cloud (keypoints, detector scores,
        keynum; no RNG), so it is computed once per cloud -- the reference recomputes it for both clouds of every pair
        (matcher.py:77-82) -- with the 5-NN search of all clouds in one segmented launch and ONE download of the neighbour lists."""
        return self._drive(self._nms_steps(clouds, keynum))

    def _nms_steps(self, clouds, keynum):
        """nms_many as a generator (see _detect_steps)."""
        from .test.matcher import NMS_sample
        sampler = NMS_sample(keynum, 5)
        todo = [c for c in clouds if keynum not in c.nms]
        need = []
        for c in todo:
            if c.keys.shape[0] < keynum:
                c.nms[keynum] = np.arange(c.keys.shape[0])
            else:
                need.append(c)
        if need:
            seg = hip.Segments([c.keys.shape[0] for c in need])
            pts = torch.cat([c.keys.float() for c in need]).contiguous()
            flat = (yield [hip.knn_search_seg(pts, seg, 5)])[0]           # every cloud's 5-NN lists (self included) in two launches
            picks = _host_pool().map(lambda q: sampler.sample_from_neighbours(need[q].det, flat[seg.host[q]:seg.host[q + 1]]), range(len(need)))
            for c, pick in zip(need, picks):
                c.nms[keynum] = pick

    # ---- per pair ------------------------------------------------------------------------------------------
    def sample(self, c0, c1, keynum, seed=None):
        """Keypoint sampling; consumes the numpy generator exactly like test/matcher.py:75-88.  seed=None: the process-global generator (the
        reference's single stream); a seed: a generator of this pair's own, `RandomState(seed)` -- the same legacy MT19937 stream that
        `np.random.seed(seed)` would start, without touching the process-global state (so pairs can be prepared on several host threads)."""
        rng = np.random if seed is None else np.random.RandomState(int(seed) % (2 ** 32))
        n0, n1 = c0.before.shape[0], c1.before.shape[0]
        if self.cfg.RD:
            # NMS sampling is a pure function of the cloud (keypoints, detector scores, keynum; no RNG): the reference recomputes
            # it for both clouds of every pair (matcher.py:77-82), here it is computed once per cloud and reused
            self.nms_many([c0, c1], keynum)
            s0, s1 = c0.nms[keynum], c1.nms[keynum]
        else:
            s0 = np.arange(n0); s1 = np.arange(n1)
            rng.shuffle(s0); rng.shuffle(s1)
            s0 = s0[0:keynum]; s1 = s1[0:keynum]
        return s0, s1

    @staticmethod
    def _match_rm_finish(s0, s1, m0, sc0):
        """matches0 / matching_scores0 (host) -> (matches [M,2] int64 host in cloud coordinates, scores f32); test/matcher.py:198-206."""
        valid = np.nonzero(m0 >= 0)[0]
        if valid.shape[0] < 3:                       # the reference crashes here; same documented divergence as test/matcher.py:
            # the single dummy correspondence (1, 1) of the SAMPLED lists, mapped to cloud rows like every other match
            return np.array([[np.asarray(s0, np.int64)[1], np.asarray(s1, np.int64)[1]]], np.int64), np.ones(1, np.float32)
        return np.stack([np.asarray(s0, np.int64)[m0[valid]], np.asarray(s1, np.int64)[valid]], 1), sc0[valid]

    def match_rm_many(self, jobs, max_points=None):
        """Rotation-coherence matcher of several pairs: jobs [(c0, c1, s0, s1)] -> [(matches [M,2] int64 device, scores f32 host)].
        One pair's kernels work on 2500 keypoints and leave most of the chip idle (and cost ~450 launches), so the sampled points of
        many pairs are stacked and the network runs ONCE per group of pairs with segmented per-pair operations (Match_ot.match_stacked;
        bitwise the per-pair forward()); one synchronisation, one download of the read-outs and one upload of the match lists per call.
        The batch carries cloud 1 as the source (feats0/keys0) and cloud 0 as the target (test/matcher.py:192-197)."""
        return self._drive(self._match_rm_steps(jobs, max_points))

    def _match_rm_steps(self, jobs, max_points=None):
        """match_rm_many as a generator (see _detect_steps): the read-outs of all groups come to the host in one (asynchronous) download."""
        if not jobs:
            return []
        max_points = self.rm_max_points if max_points is None else max_points
        # ... capped by what fits a quarter of the free device memory: per pair the Sinkhorn workspace (roreg_sinkhorn_batch3_workspace_size: no
        # coupling matrices in the default mode) plus ~60 KB of neighbourhood activations per point (k = 16 neighbours x <= 120 channels, a few tensors)
        longest = max(max(len(s0), len(s1)) for _, _, s0, s1 in jobs)
        ot_mode = (2 if hip.OT_COOP else 1) if hip.OT_RECOMPUTE else 0      # (the workspace of the mode that will actually run: with ROREG_OT_RECOMPUTE=0 a pair
        per_pair = 4 * hip.lib().roreg_sinkhorn_batch3_workspace_size(1, longest, longest, longest, longest, ot_mode, 0) + 60_000 * 2 * longest     #  keeps two coupling matrices)
        max_points = max(longest, min(max_points, int(0.25 * torch.cuda.mem_get_info()[0] / per_pair) * longest))
        flat = np.concatenate([np.ascontiguousarray(x, np.int64) for _, _, s0, s1 in jobs for x in (s0, s1)])
        flat_dev = hip.upload(flat)                                         # ONE upload of all sample lists
        rows, o = [], 0
        for _, _, s0, s1 in jobs:
            rows.append((flat_dev[o:o + len(s0)], flat_dev[o + len(s0):o + len(s0) + len(s1)])); o += len(s0) + len(s1)
        issued = []
        for i, j in even_groups([max(len(s0), len(s1)) for _, _, s0, s1 in jobs], max_points):
            seg_s = hip.Segments([len(jobs[q][3]) for q in range(i, j)]); seg_t = hip.Segments([len(jobs[q][2]) for q in range(i, j)])
            se = torch.empty((seg_s.total, 32, 60), dtype=self.feat_dtype, device='cuda'); te = torch.empty((seg_t.total, 32, 60), dtype=self.feat_dtype, device='cuda')
            sk = torch.empty((seg_s.total, 3), dtype=torch.float64, device='cuda'); tk = torch.empty((seg_t.total, 3), dtype=torch.float64, device='cuda')
            gf, gk = [], []                                                 # the group's sample gathers: two launches (features, keypoints) instead of four per pair
            for q in range(i, j):
                c0, c1 = jobs[q][0], jobs[q][1]
                d0, d1 = rows[q]
                a, b = seg_s.host[q - i], seg_t.host[q - i]
                gf += [(c1.eqv, d1, se[a:a + d1.shape[0]]), (c0.eqv, d0, te[b:b + d0.shape[0]])]
                gk += [(c1.keys, d1, sk[a:a + d1.shape[0]]), (c0.keys, d0, tk[b:b + d0.shape[0]])]
            hip.gather_rows_batch(gf); hip.gather_rows_batch(gk)
            with torch.no_grad():
                issued += self.rm.match_stacked(se.float(), te.float(), sk.float(), tk.float(), seg_s, seg_t)   # (bf16 rows: lossless up-cast)
        m0_all, sc_all = yield [torch.cat([m for m, _ in issued]), torch.cat([s for _, s in issued])]      # the one sync of the matcher stage
        out, o = [], 0
        for (c0, c1, s0, s1), (m, _) in zip(jobs, issued):
            n = int(m.shape[0])
            out.append(self._match_rm_finish(s0, s1, m0_all[o:o + n], sc_all[o:o + n])); o += n
        flat = hip.upload(np.concatenate([m.reshape(-1) for m, _ in out]))
        res, o = [], 0
        for m, sc in out:
            res.append((flat[o:o + m.size].view(-1, 2), sc)); o += m.size
        return res

    def match_rm(self, c0, c1, s0, s1):
        """Rotation-coherence matcher on the sampled keypoints (test/matcher.py:187-206) -> (matches [M,2] int64 device in cloud
        coordinates, scores float32 host)."""
        return self.match_rm_many([(c0, c1, s0, s1)])[0]

    def match_mutual(self, c0, c1, s0, s1):
        """-> (match buffer [m,2] int64 device, count int32[1] device); test/matcher.py:90-107."""
        d0 = torch.from_numpy(np.ascontiguousarray(s0, np.int64)).cuda()
        d1 = torch.from_numpy(np.ascontiguousarray(s1, np.int64)).cuda()
        buf, cnt = hip.mutual_match_batch([(c0.inv, c1.inv, d0, d1)])
        return buf[0], cnt

    def local_transforms_many(self, items, max_rows=None):
        """Des2R + ET + assembly for several pairs: per group of pairs, 2 launches (Des2R, ET input assembly), ONE pass of the ET
        network and 1 launch (quaternion -> transform).  items: [(c0, c1, matches [M,2], sel)] with sel = device int64 rows of
        `matches` to evaluate or None (= all M).  -> [(dr [n], Trans [n,3,4])] per item."""
        max_rows = self.lt_rows if max_rows is None else max_rows
        out = [None] * len(items)
        sizes = [int(it[3].shape[0]) if it[3] is not None else int(it[2].shape[0]) for it in items]
        i = 0
        while i < len(items):
            j, rows = i, 0
            while j < len(items) and (j == i or rows + sizes[j] <= max_rows):
                rows += sizes[j]; j += 1
            batch = hip.LtBatch([(c0.before, c1.before, c0.eqv, c1.eqv, c0.keys, c1.keys, m, sel, c0.eqv_ft, c1.eqv_ft) for c0, c1, m, sel in items[i:j]])
            bn = self.et.conv_init_bn()
            if bn is not None:
                dr_all, x_all, xb = batch.prepare(bound_bn=bn)          # the rows' block-scale bound comes with the assembly
            else:
                (dr_all, x_all), xb = batch.prepare(), None
            with torch.no_grad():
                q_all = self.et.trunk_and_head(x_all, x_bound=xb) if rows else torch.empty((0, 4), dtype=torch.float32, device='cuda')
            del x_all
            T_all = batch.finish(q_all, dr_all)
            for q, (o, n) in zip(range(i, j), batch.offsets):
                out[q] = (dr_all[o:o + n], T_all[o:o + n])
            i = j
        return out

    def local_transforms(self, c0, c1, matches):
        """Des2R + ET + assembly: matches [M,2] int64 device -> (dr_index [M], Trans [M,3,4] f64)."""
        rows0 = matches[:, 0].contiguous(); rows1 = matches[:, 1].contiguous()
        dr = hip.des2r(c1.eqv, c0.eqv, rows1=rows1, rows0=rows0, coefs1=c1.eqv_ft, coefs0=c0.eqv_ft)
        x = hip.et_gather(c0.before, c1.before, c0.eqv, c1.eqv, dr, rows0=rows0, rows1=rows1)
        with torch.no_grad():
            q = self.et.trunk_and_head(x)
        Trans = hip.quat_to_trans(q, dr, c0.keys, c1.keys, rows0=rows0, rows1=rows1)
        return dr, Trans, rows0, rows1

    def ransac(self, c0, c1, rows0, rows1, Trans, scores, hyp_rows, w_f32=False):
        """One-shot RANSAC + two refinements; everything stays on the device.  -> (T [4,4], best int32[1])."""
        k0 = hip.gather_rows_f64(c0.keys, rows0); k1 = hip.gather_rows_f64(c1.keys, rows1)
        ird = float(self.cfg.ransac_ird)
        _, best, _ = hip.ransac_score(k0, k1, scores, Trans, ird, hyp_rows=hyp_rows, w_f32=w_f32)
        T1, st1 = hip.refine(k0, k1, scores, ird * 2.0, Trans=Trans, hyp_rows=hyp_rows, best=best, want_stats=True, w_f32=w_f32)
        T2, st2 = hip.refine(k0, k1, scores, ird, T_in=T1, want_stats=True, w_f32=w_f32)
        return T2, best, (k0, k1, st1, st2)

    def _yohoo_tasks(self, full, all_scores, max_iter, all_local_transforms, pair_seeds=None, writer=None, pair_ids=None):
        """Estimator-tail tasks of the one-shot estimator (test/estimator.py:405-436)."""
        # One-shot RANSAC only ever reads the local transforms of the (<= max_iter) hypotheses it draws
        # (estimator.py:423-425), and that draw depends on M (and, with --RM, on the scores) alone, so the hypothesis order is drawn
        # first (same global-RNG calls in the same order as the reference) and Des2R + ET run on the selected correspondences only.
        # The registration result is identical; the reference computes all M local transforms because its stages are coupled through
        # Trans_pre files.  all_local_transforms=True evaluates every correspondence like the reference does.
        hyps = []
        rows_of = []
        for (c0, c1, matches), sc in zip(full, all_scores):
            rows = None                                                     # None = arange(M)
            if self.cfg.RM:                                                 # hypotheses only from the best-scored matches (:415-421)
                num = max(sc.shape[0] * self.cfg.match_n, 10) if self.cfg.match_n < 0.999 else self.cfg.match_n
                rows = np.argsort(sc)[-int(num):]
            rows_of.append(rows)
        n_of = [int(m.shape[0]) if r is None else int(r.shape[0]) for (_, _, m), r in zip(full, rows_of)]
        if pair_seeds is not None and len(full):
            # one threaded host call for every pair's `seed(pair seed + 1); shuffle(arange(n))[:max_iter]` (hip.mt_shuffle_prefix)
            drawn = hip.mt_shuffle_prefix(np.asarray(pair_seeds, np.int64) + 1, np.array(n_of, np.int32)[:, None], max_iter)[:, 0]
            for q, (rows, n) in enumerate(zip(rows_of, n_of)):
                index = drawn[q, :min(max_iter, n)]                         # (a contiguous int64 row view: no per-pair copy)
                hyps.append(index if rows is None else np.ascontiguousarray(rows[index], np.int64))
        else:
            drawn = hip.global_stream_shuffle_prefix(n_of, max_iter) if len(full) else None      # estimator.py:423-424 on the reference's single global stream, in C
            for q, (rows, n) in enumerate(zip(rows_of, n_of)):
                if drawn is not None:
                    index = drawn[q, :min(max_iter, n)]
                    hyps.append(index if rows is None else np.ascontiguousarray(rows[index], np.int64))
                    continue
                index = np.arange(n)
                np.random.shuffle(index)
                hyps.append(np.ascontiguousarray((index if rows is None else rows[index])[0:max_iter], np.int64))
        hyp_flat = hip.upload(np.concatenate(hyps) if hyps else np.zeros(0, np.int64))               # ONE upload of all hypothesis lists
        hyp_dev, o = [], 0
        for h in hyps:
            hyp_dev.append(hyp_flat[o:o + h.shape[0]]); o += h.shape[0]
        items = [(c0, c1, m, None if all_local_transforms else h) for (c0, c1, m), h in zip(full, hyp_dev)]
        lts = self.local_transforms_many(items)
        if writer is not None:                                             # (all_local_transforms is on: complete DR_index / Trans_pre files)
            writer.save_many([f'DR_index/{a}-{b}' for a, b in pair_ids], [dr for dr, _ in lts])
            writer.save_many([f'Trans_pre/{a}-{b}' for a, b in pair_ids], [T for _, T in lts])
        # the estimator tail of every pair in five launches (gather, score, first-best, refine x2)
        rt, w_all = [], []
        have = [sc is not None for sc in all_scores]
        w_flat = hip.upload(np.concatenate([sc.astype(np.float64) for sc in all_scores if sc is not None])) if any(have) else None   # ONE upload
        o = 0
        for (c0, c1, matches), h, sc, (dr, Trans) in zip(full, hyp_dev, all_scores, lts):
            hyp = h if all_local_transforms else None                                      # else Trans is already in hypothesis order
            w = None                                                                       # None = ones(M)  (matcher.py:109)
            if sc is not None:
                w = w_flat[o:o + sc.shape[0]]; o += sc.shape[0]
            rt.append((c0.keys, c1.keys, matches, w, Trans, hyp)); w_all.append(w)
        return rt, w_all

    def _yohoc_tasks(self, full, all_scores, max_iter, pair_seeds=None, writer=None, pair_ids=None):
        """Estimator-tail tasks of the rotation-bin RANSAC (test/estimator.py:173-241): coarse rotation of every correspondence of every
        pair in one launch (Des2R), one download, then per pair (host, the reference's order of generator calls) the bin statistics,
        the hypothesis draws and the 3-point Kabsch stack; ONE upload of all hypotheses.  -> (tasks, weights, {pair: (T, recalltime)}
        for the pairs the reference gives up on)."""
        from .test.estimator import yohoc_draws, three_point_transforms, _select_top
        batch = hip.LtBatch([(c0.eqv, c1.eqv, c0.eqv, c1.eqv, c0.keys, c1.keys, m, None, c0.eqv_ft, c1.eqv_ft) for c0, c1, m in full])
        sizes = [int(m.shape[0]) for _, _, m in full]
        # (a generator like the other stages: the coarse rotations and the match lists come to the host through run_scenes' asynchronous download,
        #  and the draws + the 3-point Kabsch stacks below -- host LAPACK, the reference's own call -- run under the other scenes' kernels)
        need_keys = [c for c0, c1, _ in full for c in (c0, c1) if c.keys_host is None]
        need_keys = list({id(c): c for c in need_keys}.values())
        got = yield [batch.des2r(), torch.cat([m.reshape(-1) for _, _, m in full]) if full else torch.zeros(0, dtype=torch.int64, device='cuda')] + [c.keys for c in need_keys]
        dr_all = np.asarray(got[0]); m_host = np.asarray(got[1]).reshape(-1, 2)
        for c, k in zip(need_keys, got[2:]):
            c.keys_host = np.array(k)
        if writer is not None:
            writer.save_many_host([f'DR_index/{a}-{b}' for a, b in pair_ids], [dr_all[off:off + n].copy() for off, n in batch.offsets])
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

        def draw(i, rng):
            """pair i's hypothesis draws -> (keys0, keys1, rows0 [H,3], rows1 [H,3]) or the reference's give-up result (estimator.py:214-216)"""
            c0, c1, _ = full[i]
            off, n = batch.offsets[i]
            pps = m_host[starts[i]:starts[i + 1]]
            sel = _select_top(all_scores[i], self.cfg.match_n) if self.cfg.RM else np.arange(n)
            idxs = yohoc_draws(dr_all[off:off + n][sel], max_iter, rng=rng)
            if idxs is None:                                               # no rotation bin with two correspondences
                return None, ((np.random if rng is None else rng).rand(4, 4), 50000)
            return (c0.keys_host, c1.keys_host, pps[sel, 0][idxs], pps[sel, 1][idxs]), None
        # the 3-point Kabsch stacks are pure functions of the draws: LAPACK releases the GIL, so the pairs run on a few host threads
        kabsch = lambda j: np.zeros((0, 3, 4)) if j is None else three_point_transforms(j[0][j[2]], j[1][j[3]])
        skipped = {}
        if pair_seeds is not None:
            # a pair's draws come from a stream of its own, RandomState(seed + 1) -- the same MT19937 stream np.random.seed(seed + 1) would start --
            # so the process-global generator is neither consumed nor left in a state that depends on the shard plan, and the pairs' draws
            # are independent of each other: draws AND Kabsch stacks of all pairs on the host pool
            sels = [_select_top(all_scores[i], self.cfg.match_n) if self.cfg.RM else None for i in range(len(full))]
            anchors = [dr_all[off:off + n] if sel is None else dr_all[off:off + n][sel] for (off, n), sel in zip(batch.offsets, sels)]
            drawn, giveup = hip.yohoc_draw_many(np.asarray(pair_seeds, np.int64) + 1, anchors, max_iter)     # one threaded host call (C: no interpreter lock)
            jobs = []
            for i, idxs in enumerate(drawn):
                if idxs is None:
                    skipped[i] = (giveup[i].copy(), 50000); jobs.append(None)
                    continue
                c0, c1, _ = full[i]
                pps = m_host[starts[i]:starts[i + 1]]
                if sels[i] is not None:
                    pps = pps[sels[i]]
                jobs.append((c0.keys_host, c1.keys_host, pps[:, 0][idxs], pps[:, 1][idxs]))
            hyps = list(_host_pool().map(kabsch, jobs))
        else:                                                              # the reference's single global stream: draws in pair order on this thread,
            futures = []                                                   # each pair's stack on the pool as soon as its draws exist
            for i in range(len(full)):
                job, gave_up = draw(i, None)
                if gave_up is not None:
                    skipped[i] = gave_up
                futures.append(_host_pool().submit(kabsch, job))
            hyps = [f.result() for f in futures]
        flat = hip.upload(np.concatenate(hyps) if hyps else np.zeros((0, 3, 4)))       # ONE upload of all hypotheses (the staging ring: no stream sync)
        have = [sc is not None for sc in all_scores]
        w_flat = hip.upload(np.concatenate([sc.astype(np.float64) for sc in all_scores if sc is not None])) if any(have) else None
        rt, w_all, o, ow = [], [], 0, 0
        for (c0, c1, matches), sc, T in zip(full, all_scores, hyps):
            w = None
            if sc is not None:
                w = w_flat[ow:ow + sc.shape[0]]; ow += sc.shape[0]
            rt.append((c0.keys, c1.keys, matches, w, flat[o:o + T.shape[0]], None)); w_all.append(w)
            o += T.shape[0]
        return rt, w_all, skipped

    # ---- whole scene -----------------------------------------------------------------------------------------
    def run_scene(self, feats, keys, pair_ids, **kw):
        """One scene, synchronously (see _scene_steps for the arguments): the scene's two host synchronisations are plain blocking downloads."""
        g = self._scene_steps(feats, keys, pair_ids, **kw)
        try:
            req = next(g)
            while True:
                req = g.send([t.cpu().numpy() for t in req])
        except StopIteration as done:
            return done.value

    def run_scenes(self, jobs):
        """Several scenes (or pair ranges), software-pipelined on the one stream: jobs = [(feats, keys, pair_ids, kwargs)] -> [[PairResult]];
        a job may also be a callable returning that tuple, evaluated when the pipeline reaches it (distributed.run_plan: the jobs that
        first have to wait for received clouds).
        A scene synchronises with the host twice -- for the match counts (the hypothesis order needs every pair's M) and for the result
        table (with --RD / --RM three more times: detector scores, NMS neighbour lists, the matcher's read-outs) -- and between a download and
        the next launch the host shuffles, builds task tables and uploads: ~5 ms per scene with the GPU
        idle when the scenes run one after the other (3 % of a step, profiles/r02_bench_gpu_idle.txt).  Here scene i + 1's extraction and
        matcher are enqueued BEFORE the host waits for scene i's counts, and scene i's estimator before it waits for scene i - 1's results;
        every download goes through a side stream ordered by an event behind its producer, so it does not wait for the kernels queued
        after it.  Needs per-pair generator streams (pair_seeds: the host-side draws then do not depend on the order scenes are
        prepared in); without them -- the reference's single global stream -- the scenes run one after the other.  Results are
        bitwise those of run_scene()."""
        import os
        materialise = lambda job: job() if callable(job) else job
        if (len(jobs) < 2 or self.phase_ms is not None or any(not callable(j) and j[3].get('pair_seeds') is None for j in jobs)
                or os.environ.get('ROREG_NO_PIPELINE')):                     # (the switch is for A/B measurements)
            return [self.run_scene(f, k, p, **kw) for f, k, p, kw in map(materialise, jobs)]
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side

        def download(tensors):
            """start the device -> pinned-host copies of `tensors` on the side stream, behind everything enqueued so far"""
            ready = torch.cuda.Event(); ready.record()
            hosts = []
            with torch.cuda.stream(side):
                side.wait_event(ready)
                for t in tensors:
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    t.record_stream(side)
                    hosts.append(h)
                done = torch.cuda.Event(); done.record(side)
            return hosts, done

        out = [None] * len(jobs)
        active = []                                                  # [index, generator, pinned hosts, event], oldest first

        def advance(slot):
            i, g, hosts, done = slot
            done.synchronize()
            try:
                req = g.send([h.numpy() for h in hosts])
            except StopIteration as fin:
                out[i] = fin.value
                return None
            return [i, g, *download(req)]

        def start(i):
            f, k, p, kw = materialise(jobs[i])
            g = self._scene_steps(f, k, p, **kw)
            try:
                return [[i, g, *download(next(g))]]
            except StopIteration as fin:
                out[i] = fin.value
                return []

        # ROREG_PIPELINE_PREFILL = extra scenes started in the first round (measured, tools/probe/prefill_ab.sh: 0 / 1 / 2 give 471-479 / 478-480 /
        # 473-478 pairs/s on the --RD --RM pipeline and 1093-1097 on the mutual one -- the queue does not run dry at the start of a step; default 0)
        prefill = int(os.environ.get('ROREG_PIPELINE_PREFILL', 0))
        nxt = 0
        while nxt < len(jobs):
            older = active
            newest = []
            for _ in range(1 + (prefill if nxt == 0 else 0)):        # the next scene's first stage is enqueued ...
                if nxt < len(jobs):
                    newest += start(nxt); nxt += 1
            # ... before the host waits for anything of the scenes already in flight, each of which then moves one stage on (oldest first)
            active = [n for n in (advance(slot) for slot in older) if n is not None] + newest
        while active:
            active = [n for n in (advance(slot) for slot in active) if n is not None]
        return out

    def _scene_steps(self, feats, keys, pair_ids, keynum=None, max_iter=None, keep_matches=False, all_local_transforms=False, pair_seeds=None,
                     writer=None, ready=None, host_svd=False):
        """Generator behind run_scene / run_scenes: yields the device tensors it needs on the host at each of the scene's two synchronisation
        points and is sent their numpy copies; returns [PairResult].
        feats/keys: dict or list indexed by int(pc_id); pair_ids: list of (id0,id1) strings.
        pair_seeds: optional one integer per pair -- the pair's keypoint sampling draws from a generator stream of its own,
        RandomState(seed) (the stream np.random.seed(seed) would start), and its hypothesis draws from RandomState(seed + 1), so a pair's
        result is a function of the pair alone (whatever other pairs this call processes: the multi-GPU driver's rank-count
        independence) and the process-global generator is left untouched.  None = the reference's single global stream.
        writer: optional StageFileWriter -- the inter-stage files of every pair (matches, scores, DR_index, Trans_pre) are written
        asynchronously, byte for byte what the file-coupled stage classes write; this turns all_local_transforms on (the files hold
        every correspondence's local transform, like the reference's).
        ready: optional {int cloud id: CloudState} of clouds that need no extraction (received from their owner rank, or extracted for
        an earlier pair range of the same scene); the clouds this call extracts are added to it.
        host_svd: close BOTH refinements of every pair with the reference's own LAPACK call on the host (one more launch + download per scene),
        like the file-coupled estimator classes do -- the result files are then theirs bit for bit; default: the device's 3x3 Jacobi SVD
        (<= 1e-10 from LAPACK's), host LAPACK only for rank-deficient covariances.
        Returns [PairResult]."""
        if writer is not None:
            all_local_transforms = True
        keynum = self.cfg.keynum if keynum is None else keynum
        max_iter = self.cfg.max_iter if max_iter is None else max_iter
        import time
        if self.phase_ms is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        used = sorted({int(i) for p in pair_ids for i in p})
        have = {} if ready is None else ready
        todo = [i for i in used if i not in have]
        save_eqv = None
        if writer is not None:
            writer.precreate_pairs(pair_ids, trans_pre=getattr(self.cfg, 'ET', 'yohoo') != 'yohoc')
        if writer is not None and writer.clouds_dir is not None:         # the extractor's file contract: float32 whatever the storage type
            def save_eqv(qs, cs):
                for q, c in zip(qs, cs):
                    writer.save_path(f'{writer.clouds_dir}/YOHO_Output_Group_feature/{todo[q]}.npy', c.eqv.float())
        fresh = dict(zip(todo, self.extract_many(_LazySeq(feats, todo), [keys[i] for i in todo], on_batch=save_eqv)))
        clouds = {i: (have[i] if i in have else fresh[i]) for i in used}
        if ready is not None:
            ready.update(fresh)
        t0 = self._mark('extract', t0)
        if self.cfg.RD:
            unscored = [i for i in used if clouds[i].det is None]
            yield from self._detect_steps([clouds[i] for i in used])
            if writer is not None and writer.clouds_dir is not None:
                for i in unscored:
                    writer.save_path(f'{writer.clouds_dir}/det_score/{i}.npy', clouds[i].det)
            t0 = self._mark('detect', t0)
            yield from self._nms_steps([clouds[i] for i in used], keynum)
            t0 = self._mark('nms', t0)
        # stage 3: all pairs
        full, all_scores = [], []
        if self.cfg.RM:
            jobs = []
            for q, (a, b) in enumerate(pair_ids):
                c0, c1 = clouds[int(a)], clouds[int(b)]
                jobs.append((c0, c1) + tuple(self.sample(c0, c1, keynum, None if pair_seeds is None else pair_seeds[q])))
            for (c0, c1, _, _), (m, sc) in zip(jobs, (yield from self._match_rm_steps(jobs))):
                full.append((c0, c1, m)); all_scores.append(sc)
            counts = np.array([m.shape[0] for _, _, m in full])
        else:
            # every pair's sampling first (host RNG in the reference's order), ONE upload of all row lists, then the whole
            # matcher stage in three launches
            if pair_seeds is not None and not self.cfg.RD and len(pair_ids):
                # every pair has a generator stream of its own: all of the scene's seeded shuffles in one threaded host call (the same
                # MT19937 streams and Fisher-Yates steps as sample()'s RandomState(seed).shuffle, replayed in C: hip.mt_shuffle_prefix)
                sizes = np.array([[clouds[int(a)].before.shape[0], clouds[int(b)].before.shape[0]] for a, b in pair_ids], np.int32)
                drawn = hip.mt_shuffle_prefix(pair_seeds, sizes, keynum)
                samples = [(drawn[q, 0, :min(keynum, int(sizes[q, 0]))], drawn[q, 1, :min(keynum, int(sizes[q, 1]))]) for q in range(len(pair_ids))]
            else:
                drawn = None
                if pair_seeds is None and not self.cfg.RD and len(pair_ids):
                    # the reference's single process-global stream, consumed in the reference's order (two shuffles per pair), replayed in C:
                    # numpy's generator is left where np.random.shuffle would have left it (hip.global_stream_shuffle_prefix)
                    sizes = np.array([[clouds[int(a)].before.shape[0], clouds[int(b)].before.shape[0]] for a, b in pair_ids], np.int32)
                    drawn = hip.global_stream_shuffle_prefix(sizes.reshape(-1), keynum)
                if drawn is not None:
                    drawn = drawn.reshape(len(pair_ids), 2, -1)
                    samples = [(drawn[q, 0, :min(keynum, int(sizes[q, 0]))], drawn[q, 1, :min(keynum, int(sizes[q, 1]))]) for q in range(len(pair_ids))]
                else:
                    samples = [self.sample(clouds[int(a)], clouds[int(b)], keynum, None if pair_seeds is None else pair_seeds[q])
                               for q, (a, b) in enumerate(pair_ids)]                                # host; runs under the extractor's kernels
            flat = np.concatenate([np.ascontiguousarray(x, np.int64) for s in samples for x in s]) if samples else np.zeros(0, np.int64)
            flat_dev = hip.upload(flat)                   # does not wait for the extractor's kernels: the task table is built under them
            tasks, o = [], 0
            for (a, b), (s0, s1) in zip(pair_ids, samples):
                c0, c1 = clouds[int(a)], clouds[int(b)]
                d0 = flat_dev[o:o + len(s0)]; o += len(s0)
                d1 = flat_dev[o:o + len(s1)]; o += len(s1)
                tasks.append((c0.inv, c1.inv, d0, d1))
            mbuf, cnt = hip.mutual_match_batch(tasks)
            counts = (yield [cnt])[0]                                        # the one sync of the matcher stage
            full = [(clouds[int(a)], clouds[int(b)], mbuf[q, :int(M)]) for q, ((a, b), M) in enumerate(zip(pair_ids, counts))]
            all_scores = [None] * len(full)
        if writer is not None:
            writer.save_many([f'{a}-{b}' for a, b in pair_ids], [m for _, _, m in full])
            writer.save_many_host([f'scores/{a}-{b}' for a, b in pair_ids], [np.ones(int(m.shape[0])) if sc is None else sc for (_, _, m), sc in zip(full, all_scores)])
        t0 = self._mark('match', t0)
        # stage 4: all pairs
        yohoc = getattr(self.cfg, 'ET', 'yohoo') == 'yohoc'
        if yohoc:
            rt, w_all, skipped = yield from self._yohoc_tasks(full, all_scores, max_iter, pair_seeds, writer, pair_ids)
        else:
            (rt, w_all), skipped = self._yohoo_tasks(full, all_scores, max_iter, all_local_transforms, pair_seeds, writer, pair_ids), {}
        t0 = self._mark('local_transforms', t0)
        ird = float(self.cfg.ransac_ird)
        f32_scores = any(sc is not None and sc.dtype == np.float32 for sc in all_scores)      # the rotation-coherence matcher's (matcher.py:210)
        best_d, T1_d, st1_d, T2_d, st2_d, rctx = hip.ransac_batch(rt, ird, w_f32=f32_scores, keep=True)
        t0 = self._mark('ransac_issue', t0)
        T_host, best_host, st_host = yield [T2_d, best_d, torch.stack([st1_d, st2_d], 1)]       # the one sync of the estimator stage
        T_host = np.array(T_host)                                           # (the driver's buffers may be read-only / pinned views)
        # The device closes each refinement with its own 3x3 SVD.  When a cross-covariance is rank-deficient
        # (<= 2 inliers: a failed registration) U V^T is not unique and the reference's value is LAPACK's; redo
        # exactly those pairs through the host-LAPACK path so engine == stages == reference in that case too.
        from .test.estimator import _kabsch_host, _dev64
        deficient = hip.stats_rank_deficient_many(st_host).any(axis=1) if len(full) else np.zeros(0, bool)       # [pairs, 2 refinements]
        redo = [i for i in range(len(full)) if i not in skipped and (deficient[i] or host_svd)]
        if redo:                                                            # their second refinements in ONE launch, one download
            T1s = np.stack([_kabsch_host(st_host[i, 0]) for i in redo])
            st_redo = (yield [hip.refine_batch(rctx, redo, T1s, ird)[1]])[0]   # (a yield point like the others: a blocking .cpu() here waited for every kernel queued behind the scene -- 60 ms per scene with --RM on pairs that fail to register)
            for i, st in zip(redo, st_redo):
                T_host[i] = _kabsch_host(st)
        local = full
        t0 = self._mark('ransac_finish', t0)
        out = []
        for i, (a, b) in enumerate(pair_ids):
            # recalltime: index of the winning hypothesis (yohoo, estimator.py:436) / its 1-based try count (yohoc, :241)
            T, rec = skipped[i] if i in skipped else (T_host[i], int(best_host[i]) + 1 if yohoc else max(int(best_host[i]), 0))
            if int(counts[i]) == 0 and i not in skipped:                  # no correspondence at all: the stage classes' result (NaN, 0)
                T = np.full((4, 4), np.nan); T[3] = [0.0, 0.0, 0.0, 1.0]; rec = 0
            out.append(PairResult(a, b, int(counts[i]), T, rec, matches=local[i][2] if keep_matches else None, scores=all_scores[i]))
        return out

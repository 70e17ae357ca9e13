"""Scene orchestration and benchmark metrics behind the reference's `yoho_evaluator` interface (test/evaluator.py:13-145).

`process_scene` is extractor -> [detector] -> matcher -> estimator.  When all four stages are the built-in classes the scene runs on the
device-resident engine (roreg_amd/engine.py) with a StageFileWriter: the tensors stay in HBM from stage to stage and every file of the
reference's contract (extractor output, detector scores, matches, scores, DR_index, Trans_pre, result .npz, pre.log) is still written --
asynchronously, byte for byte what the stage classes write from the same generator stream (ROREG_EVALUATOR=stages, or a stage object
that is not the built-in class, selects the file-coupled chain of stage.run() calls instead).  `run` then reports
feature-matching recall, inlier ratio, the PointDSC-style registration recall (RRE < 15 deg and RTE < 0.3 m, errors averaged over the
successes) and the Predator / Redwood recall of utils.RR_cal, and appends the reference's text block to {base_dir}/results.log."""
import os

import numpy as np

from ..utils import RR_cal
from ..utils.r_eval import compute_R_diff
from ..utils.utils import transform_points
from . import name2extractor, name2detector, name2matcher, name2estimator

RRE_MAX_DEG, RTE_MAX_M = 15, 0.3            # success thresholds of the pointdsc-style recall (evaluator.py:81)


def _scenes(datasets):
    """The dataset objects of a benchmark dict (its string entries are names, evaluator.py:93-95)."""
    return [d for d in datasets.values() if not isinstance(d, str)]


class _NpzTemplate:
    """np.savez(path, **arrays) for MANY archives of the same member names, dtypes and shapes: the archive np.savez itself produces for the
    first set of arrays is kept as a template, and every further one is that template with the members' data bytes and the two copies of each
    CRC-32 (local header, central directory) replaced -- ~10 us instead of the ~250 us of interpreter time np.savez holds the lock for, per
    result file of a scene (test/estimator.py:436-441 writes one per pair).  np.load() of the result equals np.load() of np.savez's file
    (member order, .npy headers, stored uncompressed); only the members' timestamps are the template's."""

    def __init__(self, arrays):
        import io
        import struct
        import zipfile
        buf = io.BytesIO()
        np.savez(buf, **arrays)
        self.base = buf.getvalue()
        self.key = self.signature(arrays)
        self.slots = []                                              # (name, data offset, data length, local CRC offset, central CRC offset)
        zf = zipfile.ZipFile(io.BytesIO(self.base))
        infos = zf.infolist()
        central, pos = {}, zf.start_dir
        for _ in infos:                                              # central directory records: signature, ..., CRC at +16, name / extra / comment lengths at +28
            assert self.base[pos:pos + 4] == b'PK\x01\x02'
            n, e, c = struct.unpack('<HHH', self.base[pos + 28:pos + 34])
            central[self.base[pos + 46:pos + 46 + n].decode()] = pos + 16
            pos += 46 + n + e + c
        for info in infos:
            ho = info.header_offset
            assert self.base[ho:ho + 4] == b'PK\x03\x04' and not info.flag_bits & 8 and info.compress_type == 0
            n, e = struct.unpack('<HH', self.base[ho + 26:ho + 30])
            a = np.asanyarray(arrays[info.filename[:-4]])
            member = info.file_size
            assert member >= a.nbytes
            self.slots.append((info.filename[:-4], ho + 30 + n + e, member, a.nbytes, ho + 14, central[info.filename]))

    @staticmethod
    def signature(arrays):
        return tuple((k, np.asanyarray(v).dtype.str, np.asanyarray(v).shape) for k, v in arrays.items())

    def fill(self, arrays):
        """-> the archive's bytes for `arrays` (same signature as the template's)"""
        import struct
        import zlib
        out = bytearray(self.base)
        for name, off, member, nbytes, crc_local, crc_central in self.slots:
            a = np.ascontiguousarray(arrays[name])
            out[off + member - nbytes:off + member] = a.tobytes()    # (the .npy header in front of it depends on dtype and shape only)
            crc = struct.pack('<I', zlib.crc32(bytes(out[off:off + member])) & 0xffffffff)
            out[crc_local:crc_local + 4] = crc
            out[crc_central:crc_central + 4] = crc
        return bytes(out)


class yoho_evaluator:
    def __init__(self, cfg):
        self.cfg = cfg
        self.GF, self.ET = cfg.GF, cfg.ET
        self.keynum, self.max_iter = cfg.keynum, cfg.max_iter
        # the flags become the stage names that label the results.log entry (evaluator.py:24-37)
        self.RD = 'yoho_det' if cfg.RD else 'nodet'
        self.RM = 'yoho_mat' if cfg.RM else 'matmul'
        self.extractor = name2extractor[self.GF](cfg)
        self.detector = name2detector[self.RD](cfg) if cfg.RD else None
        self.matcher = name2matcher[self.RM](cfg)
        self.estimator = name2estimator[self.ET](cfg)

    # ---- stages ---------------------------------------------------------------------------------------------
    def process_scene(self, dataset):
        if self._engine_route():
            return self._process_scene_engine(dataset)
        stages = [(self.extractor, ()), (self.detector, ()), (self.matcher, (self.keynum,)), (self.estimator, (self.keynum, self.max_iter))]
        for stage, args in stages:
            if stage is not None:
                stage.run(dataset, *args)

    def _engine_route(self):
        """True when the four stage objects are exactly the built-in classes (a subclass or a replaced stage keeps the run() chain) and
        ROREG_EVALUATOR is not 'stages'."""
        from .extractor import yoho_des
        from .detector import yoho_det
        from .matcher import mutual, yoho_mat
        from .estimator import yohoo, yohoc
        if os.environ.get('ROREG_EVALUATOR', 'engine') == 'stages':
            return False
        return (type(self.extractor) is yoho_des and (self.detector is None or type(self.detector) is yoho_det)
                and type(self.matcher) in (mutual, yoho_mat) and type(self.estimator) in (yohoo, yohoc))

    def _engine(self):
        """The device-resident engine over the stage objects' own networks (their checkpoints are loaded where the stage classes load them,
        with the same errors: test/extractor.py:26-31, test/estimator.py:289)."""
        if getattr(self, '_eng', None) is None:
            from ..engine import RegistrationEngine
            self.extractor._load_model()
            self.extractor.network.eval()
            et = None
            if self.ET == 'yohoo':
                self.estimator.localT_extractor._load_model()
                et = self.estimator.localT_extractor.network.eval()
            rd = self.detector.network.eval() if self.detector is not None else None
            rm = self.matcher.network.eval() if self.cfg.RM else None
            self._eng = RegistrationEngine(self.cfg, self.extractor.network, et, rd_net=rd, rm_net=rm)
        return self._eng

    def _process_scene_engine(self, dataset):
        """One scene on the engine + StageFileWriter.  Same file contract and skip rules as the stage chain: an extractor output or detector
        score file that exists is used, not recomputed (test/extractor.py:47-49, test/detector.py:37-39); matcher and estimator files are
        always rewritten; the process-global numpy generator is consumed in the reference's order (two shuffles per pair in the matcher
        without --RD, then one per pair in the estimator)."""
        from .. import hip
        from ..engine import StageFileWriter
        from . import _cache
        from ._files import SceneFiles
        from .estimator import pre_log_entry
        import time
        cfg = self.cfg
        t_start = time.perf_counter()
        eng = self._engine()
        files = SceneFiles(cfg, dataset, self.keynum)
        ft = _cache.feat_dtype(cfg)
        eng.feat_dtype = ft
        ids = [int(pc) for pc in dataset.pc_ids]
        used = sorted({int(i) for p in dataset.pair_ids for i in p})
        print(f'Registering {len(dataset.pair_ids)} pairs of {dataset.name} on the device-resident engine (stage files are written asynchronously)')

        class Inputs:
            """{cloud id: [N,32,60] float32 DEVICE tensor}: a few threads copy the scene's input files (page cache -> pinned memory with one
            readinto(), the GIL released) and start each cloud's upload on a stream of their own, ahead of the engine; asking for a cloud
            makes the current stream wait for its upload -- so the 2.3 GB of a 60-cloud scene arrive under the extractor's kernels"""

            def __init__(self, ids):
                import torch
                from concurrent.futures import ThreadPoolExecutor
                self.torch = torch
                self.pool = ThreadPoolExecutor(max(1, int(os.environ.get('ROREG_LOADER_THREADS', 8))))
                self.stream = hip.named_stream('input_loader')
                self.keep = []                                   # (the pinned buffers go back to the pool when the scene is done)
                self.read_done = []

                def read(i):
                    # a float32 C-ordered .npy (what testset.py writes) goes from the page cache into the pinned buffer with ONE readinto();
                    # anything else through numpy (a memory-mapped copy takes a page fault per 4 KB: ~1 GB/s per thread)
                    path = files.input_feature(int(i))
                    dst = None
                    with open(path, 'rb') as f:
                        version = np.lib.format.read_magic(f)
                        shape, fortran, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
                        if dtype == np.float32 and not fortran:
                            pooled = hip.pinned_pool.acquire(4 * int(np.prod(shape)))
                            dst = pooled[:4 * int(np.prod(shape))].view(torch.float32).reshape(shape)
                            buf = memoryview(dst.numpy()).cast('B')
                            got = 0
                            while got < len(buf):
                                n = f.readinto(buf[got:])
                                if not n:
                                    raise IOError(f'{path}: truncated')
                                got += n
                    if dst is None:
                        src = np.load(path, mmap_mode='r')
                        pooled = hip.pinned_pool.acquire(4 * int(np.prod(src.shape)))
                        dst = pooled[:4 * int(np.prod(src.shape))].view(torch.float32).reshape(src.shape)
                        np.copyto(dst.numpy(), src, casting='same_kind')
                    with torch.cuda.stream(self.stream):
                        dev = dst.to('cuda', non_blocking=True)
                        done = torch.cuda.Event(); done.record(self.stream)
                    self.keep.append(pooled)
                    self.read_done.append(time.perf_counter())
                    return dev, done
                self.jobs = {i: self.pool.submit(read, i) for i in ids}

            def __getitem__(self, i):
                dev, done = self.jobs[int(i)].result()
                cur = self.torch.cuda.current_stream()
                cur.wait_event(done)
                dev.record_stream(cur)                           # (allocated on the loader's stream, used on this one)
                return dev

            def close(self):
                self.pool.shutdown(wait=True)
                self.stream.synchronize()                        # (every upload has run: the buffers may be overwritten)
                for b in self.keep:
                    hip.pinned_pool.release(b)
                self.keep.clear()
        marks = [('engine_ready', time.perf_counter() - t_start)]
        feats = Inputs(ids)
        keys = {i: dataset.get_kps(str(i)) for i in ids}
        marks.append(('keypoints_loaded', time.perf_counter() - t_start))
        writer = StageFileWriter(cfg, dataset.name, self.keynum, clouds_dir=files.clouds)
        files.make(files.result_dir(self.ET, self.max_iter))
        writer.also_precreate = [files.result(self.ET, self.max_iter, a, b) for a, b in dataset.pair_ids]
        marks.append(('writer_ready', time.perf_counter() - t_start))
        ready = {}
        for i in ids:
            if os.path.exists(files.feature(i)):                 # extracted by an earlier run: the file is the contract
                ready[i] = eng.cloud_from_eqv(feats[i], _cache.load_device(files.feature(i), ft), keys[i])
                if cfg.RD and os.path.exists(files.det_score(i)):
                    ready[i].det = np.load(files.det_score(i))
        try:
            extra = [i for i in ids if i not in used and i not in ready]       # clouds no pair touches still get their stage files
            if extra:
                for i, c in zip(extra, eng.extract_many([feats[i] for i in extra], [keys[i] for i in extra])):
                    ready[i] = c
                    writer.save_path(files.feature(i), c.eqv.float())
            unscored = [i for i in ids if i not in used and ready[i].det is None] if cfg.RD else []
            if unscored:
                eng.detect_many([ready[i] for i in unscored])
                for i in unscored:
                    writer.save_path(files.det_score(i), ready[i].det)
            res = eng.run_scene(feats, keys, dataset.pair_ids, keynum=self.keynum, max_iter=self.max_iter, writer=writer, ready=ready, host_svd=True)
            t_run = time.perf_counter()
            # the result files and pre.log (test/estimator.py:14-26, 436-441) while the writer's threads drain the stage files
            rdir = files.result_dir(self.ET, self.max_iter)

            def write_results(part):                              # (np.savez's archive per pair, from a template: _NpzTemplate)
                paths, blobs = [], []
                for r in part:
                    arrays = {'trans': r.trans}
                    if self.ET == 'yohoc' and r.recalltime == 50000:
                        arrays['center'] = np.ones([6, 3])
                    arrays['recalltime'] = r.recalltime
                    sig = _NpzTemplate.signature(arrays)
                    tpl = templates.get(sig)
                    if tpl is None:
                        tpl = templates[sig] = _NpzTemplate(arrays)
                    paths.append(files.result(self.ET, self.max_iter, r.id0, r.id1)); blobs.append(tpl.fill(arrays))
                hip.write_files(paths, blobs, n_threads=2)
            templates = {}
            write_results(res[:1])                                # (the template is made here, not raced for on the writer's threads)
            for q in range(1, len(res), 128):
                writer.submit(lambda part=res[q:q + 128]: write_results(part))
            with open(f'{rdir}/pre.log', 'w') as log:
                log.write(''.join(pre_log_entry(r.id0, r.id1, len(dataset.pc_ids), r.trans) for r in res))
            t_res = time.perf_counter()
        finally:
            writer.close()
            feats.close()
        t_end = time.perf_counter()
        marks += [('first_input_in_pinned_memory', min(feats.read_done, default=t_start) - t_start), ('last_input_in_pinned_memory', max(feats.read_done, default=t_start) - t_start),
                  ('results_on_host', t_run - t_start), ('result_files_queued', t_res - t_start), ('writer_closed', t_end - t_start)]
        for kind in sorted({e[0] for e in writer.log}):          # the writer's side: when each kind of file had landed in pinned memory / was on disk
            es = [e for e in writer.log if e[0] == kind]
            marks.append((f'{kind}: {len(es)} entries, {sum(e[1] for e in es) / 1e6:.0f} MB, first landed', min(e[2] for e in es) - t_start))
            marks.append((f'{kind}: last landed', max(e[2] for e in es) - t_start))
            marks.append((f'{kind}: last written', max(e[3] for e in es) - t_start))
        self.last_scene_timeline = marks
        self.last_scene_seconds = {'engine_until_results_on_host': t_run - t_start, 'result_files': t_res - t_run, 'waiting_for_the_stage_file_writer': t_end - t_res}

    # ---- metrics --------------------------------------------------------------------------------------------
    def _match_dir(self, dataset):
        return f'{self.cfg.output_cache_fn}/{dataset.name}/match_{self.keynum}'

    def fmr_ir_scene(self, dataset):
        """(feature matching recall, mean inlier ratio) of a scene; with the rotation-coherence matcher only the best-scored
        share `match_n` of the correspondences counts (evaluator.py:55-59)."""
        ratios = []
        for a, b in dataset.pair_ids:
            pairs = np.load(f'{self._match_dir(dataset)}/{a}-{b}.npy')
            if self.cfg.RM:
                w = np.load(f'{self._match_dir(dataset)}/scores/{a}-{b}.npy')
                keep = max(w.shape[0] * self.cfg.match_n, 10) if self.cfg.match_n < 0.999 else self.cfg.match_n
                pairs = pairs[np.argsort(w)[-int(keep):]]
            p0 = dataset.get_kps(a)[pairs[:, 0]]
            p1 = transform_points(dataset.get_kps(b)[pairs[:, 1]], dataset.get_transform(a, b))
            ratios.append(np.mean(np.sqrt(np.sum(np.square(p0 - p1), axis=-1)) < self.cfg.tau_2))
        hits = [1 if r > self.cfg.tau_1 else 0 for r in ratios]
        return np.mean(np.array(hits)), np.mean(np.array(ratios))

    def rr_scene(self, dataset):
        """(registration recall, mean RRE, mean RTE) with the errors averaged over the successful pairs only."""
        ok, e_rot, e_tra = [], [], []
        for a, b in dataset.pair_ids:
            T = np.load(f'{self._match_dir(dataset)}/{self.ET}/{self.max_iter}iters/{a}-{b}.npz')['trans']
            G = dataset.get_transform(a, b)
            dr = compute_R_diff(T[0:3, 0:3], G[0:3, 0:3])
            dt = np.sqrt(np.sum(np.square(T[0:3, -1] - G[0:3, -1])))
            good = (dr < RRE_MAX_DEG) and (dt < RTE_MAX_M)
            ok.append(1 if good else 0)
            if good:
                e_rot.append(dr); e_tra.append(dt)
        return np.mean(np.array(ok)), np.mean(np.array(e_rot)), np.mean(np.array(e_tra))

    def run(self, datasets=None):
        if datasets is None:
            from ..dataops.dataset import get_dataset_name
            datasets = get_dataset_name(self.cfg.testset, self.cfg.origin_data_dir)
        scenes = _scenes(datasets)
        for d in scenes:
            self.process_scene(d)
        per_scene = np.array([self.fmr_ir_scene(d) + self.rr_scene(d) for d in scenes])      # [scene, (fmr, ir, rr, rre, rte)]
        fmr, ir, rr, rre, rte = (np.mean(per_scene[:, c]) for c in range(5))
        whole = datasets['wholesetname']
        rr_predator = 1.0 if whole == 'demo' else RR_cal.benchmark(self.cfg, datasets, self.keynum, self.max_iter, yoho_sign=self.ET)[0]
        rows = [('feature matching recall', fmr), ('inlier ratio', ir), ('registration recall(predator)', rr_predator),
                ('rotation error(pointdsc)', rre), ('translation error(pointdsc)', rte), ('registration recall(pointdsc)', rr)]
        msg = f'{whole}-{self.GF}-{self.RD}-{self.RM}-{self.ET}-{self.keynum}keys-{self.max_iter}iters\n'
        msg += '\n'.join(f'{label:<33}: {value:.5f}' for label, value in rows)
        with open(f'{self.cfg.base_dir}/results.log', 'a') as f:
            f.write(msg + '\n')
        print(msg)
        return {'fmr': fmr, 'ir': ir, 'rr_predator': rr_predator, 'rre': rre, 'rte': rte, 'rr': rr}

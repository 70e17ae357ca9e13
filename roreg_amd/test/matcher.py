"""Stage 3: matching (mirror of test/matcher.py:11-210).

  NMS_sample(num,k).sample(keys,scores)  -- non-maximum suppression sampling on detector scores
  mutual(cfg).run(dataset,keynum)        -- mutual nearest neighbours of the invariant descriptors
  yoho_mat(cfg).run(dataset,keynum)      -- rotation-coherence matcher (Match_ot)

Outputs: match_{keynum}/{id0}-{id1}.npy [M,2] int64 (col0 -> pc0, col1 -> pc1) and scores/{id0}-{id1}.npy."""
import numpy as np
import torch
import tqdm

from .. import hip
from ..network import name2network
from ..utils.knn_search import knn_module
from ..utils.utils import to_cuda
from . import _cache
from ._files import SceneFiles
from .extractor import restore_weights


class NMS_sample():
    def __init__(self, num, k):
        '''Non-maximum suppression'''
        self.k = k
        self.num = num
        self.KNN = knn_module.KNN(self.k)

    def sample(self, keys, scores):
        if keys.shape[0] < self.num:                     # NB '<': equal sizes still go through NMS (matcher.py:19)
            return np.arange(keys.shape[0])
        keys = torch.from_numpy(keys.astype(np.float32)[None, :, :]).permute(0, 2, 1)
        d, argmin = self.KNN(keys, keys)
        argmin = argmin[0].permute(1, 0).cpu().numpy()   # N*k
        return self.sample_from_neighbours(scores, argmin)

    def sample_from_neighbours(self, scores, argmin):
        """The selection of matcher.py:24-41 given the k nearest neighbours (self included) of every keypoint, argmin [N,k]:
        keypoints that are the maximum of their neighbourhood survive; too many -> the `num` best of them; too few -> filled up with
        the best-scoring suppressed keypoints (survivors first, then the fill, each in the order numpy's argsort gives)."""
        peak = np.max(scores[argmin.reshape(-1)].reshape(-1, self.k), axis=-1)
        chosen = np.where(scores >= peak)[0]
        if chosen.shape[0] > self.num:
            w = scores[chosen]
            w = w / np.sum(w)                                   # as in the reference: the ranking is taken on the normalised scores
            chosen = chosen[np.argsort(w)[-self.num:]]
        missing = self.num - chosen.shape[0]
        if missing > 0:
            suppressed = np.where(scores < peak)[0]
            fill = suppressed[np.argsort(scores[suppressed])[-missing:]]
            chosen = np.concatenate([chosen, fill], axis=0)
        return chosen


def _sample_pair(cfg, sampler, dataset, files, id0, id1, n0, n1, keynum):
    """Keypoint sampling of one pair -> (rows of cloud 0, rows of cloud 1): NMS on the detector's rank scores with --RD, else the first
    `keynum` of a shuffle of each cloud's rows -- two shuffles per pair on the process-global generator, cloud 0 first, exactly the
    reference's consumption (matcher.py:75-88)."""
    if cfg.RD:
        return tuple(sampler.sample(dataset.get_kps(pc), np.load(files.det_score(pc))) for pc in (id0, id1))
    drawn = []
    for n in (n0, n1):
        rows = np.arange(n)
        np.random.shuffle(rows)
        drawn.append(rows[0:keynum])
    return tuple(drawn)


class mutual():
    def __init__(self, cfg):
        self.cfg = cfg
        self.KNN = knn_module.KNN(1)

    def run(self, dataset, keynum=5000):
        self.sampler = NMS_sample(keynum, 5)
        print(f'Matching the keypoints with mutual on {dataset.name}')
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make('scores')
        ft = _cache.feat_dtype(self.cfg)
        for id0, id1 in tqdm.tqdm(dataset.pair_ids):
            eqv0 = _cache.load_device(files.feature(id0), ft)            # N*32*60, HBM-resident across pairs
            eqv1 = _cache.load_device(files.feature(id1), ft)
            inv0 = hip.inv_descriptor(eqv0)                               # mean over g, / (norm + 1e-5)  (matcher.py:69-72)
            inv1 = hip.inv_descriptor(eqv1)
            sample0, sample1 = _sample_pair(self.cfg, self.sampler, dataset, files, id0, id1, eqv0.shape[0], eqv1.shape[0], keynum)
            s0 = torch.from_numpy(np.ascontiguousarray(sample0, np.int64)).cuda()
            s1 = torch.from_numpy(np.ascontiguousarray(sample1, np.int64)).cuda()
            buf, cnt = hip.mutual_match_batch([(inv0, inv1, s0, s1)])       # both NN directions + the mutual check (matcher.py:90-107)
            found = buf[0, :int(cnt.item())].cpu().numpy()
            np.save(files.matches(id0, id1), found)
            np.save(files.scores(id0, id1), np.ones(found.shape[0]))     # (float64 ones, matcher.py:109)


class yoho_mat():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['RM_test'](self.cfg)
        self.best_model_fn = f'{self.cfg.model_fn}/RM/model_best.pth'
        self.KNN = knn_module.KNN(1)
        self._load_model()

    def _load_model(self):
        restore_weights(self.network, self.best_model_fn, strict=True)

    def get_ot_match(self, batch):
        """The network on one pair's batch dict -> (pairs [M,2] = (source row, its mutual match) for the matched source rows in increasing
        order, or None when fewer than three rows are matched; their matching scores; matching_scores0; matching_scores1), all host arrays
        (matcher.py:131-150)."""
        self.network.eval()
        with torch.no_grad():
            out = self.network(to_cuda(batch))
        partner, scores0, scores1 = (out[k][0].cpu().numpy() for k in ('matches0', 'matching_scores0', 'matching_scores1'))
        matched = np.where(partner != -1)[0]
        pairs = np.stack([matched, partner[matched]], 1) if matched.shape[0] >= 3 else None
        return pairs, np.array(scores0[matched]), scores0, scores1

    def run(self, dataset, keynum=2500):
        self.sampler = NMS_sample(keynum, 5)
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make('scores')
        print(f'Matching the keypoints with rotation coherence matcher on {dataset.name}')
        # The reference runs the network pair by pair (matcher.py:187-206); one pair's 2500 points leave most of the chip idle, so the
        # sampled points of a group of pairs are stacked and go through the network in one pass with segmented per-pair operations
        # (Match_ot.match_many: bitwise the per-pair forward).  Sampling order (and its generator calls) is the reference's.
        self.network.eval()
        group, pend = int(getattr(self.cfg, 'rm_group', 16)), []
        ft = _cache.feat_dtype(self.cfg)

        def flush():
            if not pend:
                return
            with torch.no_grad():
                outs = self.network.match_many([q[2] for q in pend])
            partner_all = torch.cat([m for m, _ in outs]).cpu().numpy(); score_all = torch.cat([x for _, x in outs]).cpu().numpy()
            o = 0
            for (id0, id1, _, sample0, sample1), (m, _) in zip(pend, outs):
                n = int(m.shape[0])
                partner, sc = partner_all[o:o + n], score_all[o:o + n]; o += n
                matched = np.where(partner != -1)[0]
                if matched.shape[0] < 3:
                    # the reference crashes here (np.ones(1,2) is a TypeError, matcher.py:200-202); documented
                    # divergence: emit the single dummy correspondence it evidently intended
                    src_rows, tgt_rows, scores = np.ones(1, np.int64), np.ones(1, np.int64), np.ones(1, np.float32)
                else:
                    src_rows, tgt_rows, scores = matched, partner[matched], sc[matched]
                # the network's source side is cloud 1, its target side cloud 0: column 0 of the file indexes cloud 0 (matcher.py:204-206)
                np.save(files.matches(id0, id1), np.stack([sample0[tgt_rows], sample1[src_rows]], 1))
                np.save(files.scores(id0, id1), scores)
            pend.clear()

        for id0, id1 in tqdm.tqdm(dataset.pair_ids):
            feats0 = _cache.load_device(files.feature(id0), ft)
            feats1 = _cache.load_device(files.feature(id1), ft)
            sample0, sample1 = _sample_pair(self.cfg, self.sampler, dataset, files, id0, id1, feats0.shape[0], feats1.shape[0], keynum)
            s0 = torch.from_numpy(np.ascontiguousarray(sample0, np.int64)).cuda()
            s1 = torch.from_numpy(np.ascontiguousarray(sample1, np.int64)).cuda()
            keys0 = dataset.get_kps(id0)[sample0]
            keys1 = dataset.get_kps(id1)[sample1]
            # NB the network's source side ('feats0/keys0') is pc1 and its target side pc0 (matcher.py:192-197)
            pend.append((id0, id1, (feats1[s1].float(), feats0[s0].float(), torch.from_numpy(keys1.astype(np.float32)).cuda(),
                                    torch.from_numpy(keys0.astype(np.float32)).cuda()), np.asarray(sample0), np.asarray(sample1)))
            if len(pend) >= group:
                flush()
        flush()

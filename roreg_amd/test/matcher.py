"""Stage 3: matching (mirror of test/matcher.py:11-210).

  NMS_sample(num,k).sample(keys,scores)  -- non-maximum suppression sampling on detector scores
  mutual(cfg).run(dataset,keynum)        -- mutual nearest neighbours of the invariant descriptors
  yoho_mat(cfg).run(dataset,keynum)      -- rotation-coherence matcher (Match_ot)

Outputs: match_{keynum}/{id0}-{id1}.npy [M,2] int64 (col0 -> pc0, col1 -> pc1) and scores/{id0}-{id1}.npy."""
import os

import numpy as np
import torch
import tqdm

from .. import hip
from ..network import name2network
from ..utils.knn_search import knn_module
from ..utils.utils import make_non_exists_dir, to_cuda, load_checkpoint
from . import _cache
from .extractor import scene_feature_name


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


def _sample_pair(cfg, sampler, dataset, datasetname, id0, id1, n0, n1, keynum):
    """Keypoint sampling of one pair; identical global-RNG consumption to matcher.py:75-88."""
    if cfg.RD:
        det_scores0 = np.load(f'{cfg.output_cache_fn}/{datasetname}/det_score/{id0}.npy')
        det_scores1 = np.load(f'{cfg.output_cache_fn}/{datasetname}/det_score/{id1}.npy')
        sample0 = sampler.sample(dataset.get_kps(id0), det_scores0)
        sample1 = sampler.sample(dataset.get_kps(id1), det_scores1)
    else:
        sample0 = np.arange(n0)
        sample1 = np.arange(n1)
        np.random.shuffle(sample0)
        np.random.shuffle(sample1)
        sample0 = sample0[0:keynum]
        sample1 = sample1[0:keynum]
    return sample0, sample1


class mutual():
    def __init__(self, cfg):
        self.cfg = cfg
        self.KNN = knn_module.KNN(1)

    def run(self, dataset, keynum=5000):
        self.sampler = NMS_sample(keynum, 5)
        print(f'Matching the keypoints with mutual on {dataset.name}')
        Save_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        make_non_exists_dir(Save_dir)
        Save_score_dir = f'{Save_dir}/scores'
        make_non_exists_dir(Save_score_dir)
        datasetname = scene_feature_name(dataset)
        Feature_dir = f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature'
        for pair in tqdm.tqdm(dataset.pair_ids):
            id0, id1 = pair
            ft = _cache.feat_dtype(self.cfg)
            eqv0 = _cache.load_device(f'{Feature_dir}/{id0}.npy', ft)   # N*32*60, HBM-resident across pairs
            eqv1 = _cache.load_device(f'{Feature_dir}/{id1}.npy', ft)
            inv0 = hip.inv_descriptor(eqv0)                               # mean over g, / (norm + 1e-5)  (matcher.py:69-72)
            inv1 = hip.inv_descriptor(eqv1)
            sample0, sample1 = _sample_pair(self.cfg, self.sampler, dataset, datasetname, id0, id1, eqv0.shape[0], eqv1.shape[0], keynum)
            s0 = torch.from_numpy(np.ascontiguousarray(sample0, np.int64)).cuda()
            s1 = torch.from_numpy(np.ascontiguousarray(sample1, np.int64)).cuda()
            buf, cnt = hip.mutual_match_batch([(inv0, inv1, s0, s1)])       # both NN directions + the mutual check (matcher.py:90-107)
            match_pps = buf[0, :int(cnt.item())].cpu().numpy()
            np.save(f'{Save_dir}/{id0}-{id1}.npy', match_pps)
            np.save(f'{Save_score_dir}/{id0}-{id1}.npy', np.ones(match_pps.shape[0]))


class yoho_mat():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['RM_test'](self.cfg)
        self.best_model_fn = f'{self.cfg.model_fn}/RM/model_best.pth'
        self.KNN = knn_module.KNN(1)
        self._load_model()

    def _load_model(self):
        if os.path.exists(self.best_model_fn):
            checkpoint = load_checkpoint(self.best_model_fn)
            self.network.load_state_dict(checkpoint['network_state_dict'], strict=True)
        else:
            raise ValueError("No model exists")

    def get_ot_match(self, batch):
        self.network.eval()
        with torch.no_grad():
            result = self.network(to_cuda(batch))
        matches0 = result['matches0'][0].cpu().numpy()
        scores = result['matching_scores0'][0].cpu().numpy()
        scores0 = scores
        scores1 = result['matching_scores1'][0].cpu().numpy()
        valid = np.where(matches0 != -1)[0]
        score_ms = scores[valid]
        if valid.shape[0] < 3:
            pairs = None
        else:
            pairs = np.stack([valid, matches0[valid]], 1)
        return pairs, np.array(score_ms), scores0, scores1

    def run(self, dataset, keynum=2500):
        self.sampler = NMS_sample(keynum, 5)
        Save_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        make_non_exists_dir(Save_dir)
        Save_score_dir = f'{Save_dir}/scores'
        make_non_exists_dir(Save_score_dir)
        datasetname = scene_feature_name(dataset)
        Feature_dir = f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature'
        print(f'Matching the keypoints with rotation coherence matcher on {dataset.name}')
        # The reference runs the network pair by pair (matcher.py:187-206); one pair's 2500 points leave most of the chip idle, so the
        # sampled points of a group of pairs are stacked and go through the network in one pass with segmented per-pair operations
        # (Match_ot.match_many: bitwise the per-pair forward).  Sampling order (and its generator calls) is the reference's.
        self.network.eval()
        group, pend = int(getattr(self.cfg, 'rm_group', 16)), []

        def flush():
            if not pend:
                return
            with torch.no_grad():
                outs = self.network.match_many([q[2] for q in pend])
            m0_all = torch.cat([m for m, _ in outs]).cpu().numpy(); sc_all = torch.cat([x for _, x in outs]).cpu().numpy()
            o = 0
            for (id0, id1, _, sample0, sample1), (m, _) in zip(pend, outs):
                n = int(m.shape[0])
                matches0, sc = m0_all[o:o + n], sc_all[o:o + n]; o += n
                valid = np.where(matches0 != -1)[0]
                if valid.shape[0] < 3:
                    # the reference crashes here (np.ones(1,2) is a TypeError, matcher.py:200-202); documented
                    # divergence: emit the single dummy correspondence it evidently intended
                    matches = np.ones((1, 2), np.int64)
                    scores = np.ones(1, np.float32)
                else:
                    matches = np.stack([valid, matches0[valid]], 1)
                    scores = sc[valid]
                matches_in_former = np.concatenate([sample0[matches[:, 1]][:, None], sample1[matches[:, 0]][:, None]], axis=1)
                np.save(f'{Save_dir}/{id0}-{id1}.npy', matches_in_former)
                np.save(f'{Save_score_dir}/{id0}-{id1}.npy', scores)
            pend.clear()

        for pair in tqdm.tqdm(dataset.pair_ids):
            id0, id1 = pair
            ft = _cache.feat_dtype(self.cfg)
            feats0 = _cache.load_device(f'{Feature_dir}/{id0}.npy', ft)
            feats1 = _cache.load_device(f'{Feature_dir}/{id1}.npy', ft)
            sample0, sample1 = _sample_pair(self.cfg, self.sampler, dataset, datasetname, id0, id1, feats0.shape[0], feats1.shape[0], keynum)
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

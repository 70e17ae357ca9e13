"""The on-disk contract between the file-coupled stages of one scene, in one place.

Everything lives under `cfg.output_cache_fn`:

    {feature scene}/{backbone}_Input_Group_feature/{pc}.npy    float32 [N,32,60]     backbone output on the 60 rotated copies (input)
    {feature scene}/YOHO_Output_Group_feature/{pc}.npy         float32 [N,32,60]     extractor
    {feature scene}/det_score/{pc}.npy                         float32 [N]           detector (rank / N)
    {scene}/match_{keynum}/{a}-{b}.npy                         int64   [M,2]         matcher (column 0 -> cloud a, column 1 -> cloud b)
    {scene}/match_{keynum}/scores/{a}-{b}.npy                  float64 ones | float32 [M]
    {scene}/match_{keynum}/DR_index/{a}-{b}.npy                int64   [M]           coarse rotation of every correspondence
    {scene}/match_{keynum}/Trans_pre/{a}-{b}.npy               float64 [M,3,4]       local transform of every correspondence
    {scene}/match_{keynum}/{estimator}/{max_iter}iters/{a}-{b}.npz   {trans [4,4] float64, recalltime}  and pre.log beside them

where {feature scene} is {scene} except that the low-overlap split '3dLomatch/x' shares the clouds of '3dmatch/x'
(test/extractor.py:38-41, test/matcher.py:54-62, test/estimator.py:96-111,330-336)."""
import numpy as np

from ..utils.utils import make_non_exists_dir
from .extractor import scene_feature_name


class SceneFiles:
    def __init__(self, cfg, dataset, keynum=None):
        root = cfg.output_cache_fn
        self.clouds = f'{root}/{scene_feature_name(dataset)}'
        self.inputs = f'{self.clouds}/{cfg.backbone}_Input_Group_feature'
        self.features = f'{self.clouds}/YOHO_Output_Group_feature'
        self.saliency = f'{self.clouds}/det_score'
        self.pairs = None if keynum is None else f'{root}/{dataset.name}/match_{keynum}'

    # ---- per cloud ----
    def input_feature(self, pc):
        return f'{self.inputs}/{pc}.npy'

    def feature(self, pc):
        return f'{self.features}/{pc}.npy'

    def det_score(self, pc):
        return f'{self.saliency}/{pc}.npy'

    # ---- per pair ----
    def _pair(self, sub, a, b):
        return f'{self.pairs}/{sub}{a}-{b}.npy'

    def matches(self, a, b):
        return self._pair('', a, b)

    def scores(self, a, b):
        return self._pair('scores/', a, b)

    def dr_index(self, a, b):
        return self._pair('DR_index/', a, b)

    def trans_pre(self, a, b):
        return self._pair('Trans_pre/', a, b)

    def result_dir(self, estimator, max_iter):
        return f'{self.pairs}/{estimator}/{max_iter}iters'

    def result(self, estimator, max_iter, a, b):
        return f'{self.result_dir(estimator, max_iter)}/{a}-{b}.npz'

    def make(self, *subdirs):
        """Create match_{keynum}/ and the named sub-directories ('scores', 'DR_index', 'Trans_pre', an estimator's result directory)."""
        make_non_exists_dir(self.pairs)
        for d in subdirs:
            make_non_exists_dir(d if d.startswith(self.pairs) else f'{self.pairs}/{d}')

    def load_matches(self, a, b):
        return np.load(self.matches(a, b))

    def load_scores(self, a, b):
        return np.load(self.scores(a, b))

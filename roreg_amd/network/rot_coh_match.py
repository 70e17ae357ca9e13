"""Match_ot: rotation-coherence matcher (network/rot_coh_match.py:323-390).  Parameter layout mirrors the
reference so RM checkpoints load; the HIP forward lands in a later milestone of this round."""
from copy import deepcopy

import torch
import torch.nn as nn


class mlp_2layer(nn.Module):
    def __init__(self, in_dim, middle_dim, out_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(in_dim, middle_dim, 1, 1), nn.InstanceNorm2d(middle_dim), nn.ReLU(),
                                 nn.Conv2d(middle_dim, out_dim, 1, 1))
        self.res_sign = in_dim != out_dim
        if self.res_sign:
            self.res = nn.Conv2d(in_dim, out_dim, 1, 1)


class Contextnorm(mlp_2layer):
    pass


class MultiHeadedAttention(nn.Module):
    def __init__(self, num_heads, d_model):
        super().__init__()
        self.dim = d_model // num_heads
        self.num_heads = num_heads
        self.merge = nn.Conv2d(d_model, d_model, kernel_size=1, stride=1)
        self.proj = nn.ModuleList([deepcopy(self.merge) for _ in range(3)])


class Cross_attention_block(nn.Module):
    def __init__(self, cross_k, s2t):
        super().__init__()
        self.k = cross_k
        self.s2t = s2t
        self.cross_attn = MultiHeadedAttention(4, 32)
        self.merge = mlp_2layer(32 * 3, 64, 32)


class Self_attention_block(nn.Module):
    def __init__(self, self_k, source):
        super().__init__()
        self.k = self_k
        self.source = source
        self.self_attn = MultiHeadedAttention(4, 32)
        self.pos_en = mlp_2layer(3, 64, 32)
        self.ambiguity = Contextnorm(120, 128, 32)
        self.val_en = mlp_2layer(32 * 3, 64, 32)
        self.merge = mlp_2layer(32 * 3, 64, 32)


class Merge_info_block(nn.Module):
    def __init__(self, self_k, cross_k):
        super().__init__()
        self.cross_graph_s2t = Cross_attention_block(cross_k, s2t=True)
        self.self_graph_s = Self_attention_block(self_k, source=True)
        self.cross_graph_t2s = Cross_attention_block(cross_k, s2t=False)
        self.self_graph_t = Self_attention_block(self_k, source=False)


class Graph_enhance_net(nn.Module):
    def __init__(self):
        super().__init__()
        self.merge_blocks = nn.ModuleList([Merge_info_block(16, 16), Merge_info_block(8, 8)])


class sinkhorn_ot(nn.Module):
    def __init__(self, origin_bin, iters):
        super().__init__()
        self.iters = iters
        self.register_parameter('bin_score', torch.nn.Parameter(torch.tensor(origin_bin)))


class Match_ot(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.coor_norm_step = 0.025
        self.Graph = Graph_enhance_net()
        self.final_mlp = mlp_2layer(64, 64, 32)
        self.ot_layer = sinkhorn_ot(0.2, 100)

    def forward(self, batch):
        raise NotImplementedError('Match_ot HIP forward: not built yet in this round (DESIGN.md, scope row A6)')

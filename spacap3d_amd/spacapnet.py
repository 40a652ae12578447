"""``SpaCapNet`` -- counterpart of the reference's ``models/SpaCapNet.py`` (same constructor, same data_dict
keys): backbone -> voting -> L2-normalise votes -> proposal -> transformer captioner."""
import torch
import torch.nn as nn

from .detector import Pointnet2Backbone, ProposalModule, VotingModule
from .layout import ChannelMajorOf, point_major_of
from .transformer_captioner import TransformerDecoderModel


class SpaCapNet(nn.Module):
    def __init__(self, num_class, vocabulary, num_heading_bin, num_size_cluster, mean_size_arr,
                 input_feature_dim=0, num_proposal=256, vote_factor=1, sampling="vote_fps", no_caption=False,
                 N=6, h=8, d_model=128, d_ff=2048, transformer_dropout=0.1, bn_momentum=0.1, src_pos_type=None,
                 use_transformer_encoder=False, early_guide=False, check_relation=False, store_attn_all=False):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        assert mean_size_arr.shape[0] == num_size_cluster
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.vote_factor = vote_factor
        self.sampling = sampling
        self.no_caption = no_caption
        self.backbone_net = Pointnet2Backbone(input_feature_dim=input_feature_dim)
        self.vgen = VotingModule(vote_factor, 256)
        self.proposal = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                                       sampling, size_decoded=(src_pos_type == "loc"))
        if not no_caption:
            self.caption = TransformerDecoderModel(vocabulary, N, h, d_model, d_ff, transformer_dropout,
                                                   bn_momentum=bn_momentum, src_pos_type=src_pos_type,
                                                   use_transformer_encoder=use_transformer_encoder,
                                                   early_guide=early_guide, check_relation=check_relation,
                                                   store_attn_all=store_attn_all)

    def forward(self, data_dict, is_eval=False, after_proposal=None):
        """``after_proposal``: optional callable(data_dict) invoked between the proposal module and the captioner
        (the training engine starts the detection losses there, on a side stream)."""
        data_dict = self.backbone_net(data_dict)
        xyz, features = data_dict["fp2_xyz"], data_dict["fp2_features"]
        data_dict["seed_inds"] = data_dict["fp2_inds"]
        data_dict["seed_xyz"] = xyz
        data_dict["seed_features"] = features
        xyz, features = self.vgen(xyz, features)
        pm = point_major_of(features)
        if pm is not None:   # same L2 normalisation over the channels, on the point-major tensor
            from .backend import ops
            f = getattr(ops(), "l2norm_rows", None) if (pm.is_cuda and pm.shape[-1] % 4 == 0) else None
            pm = f(pm) if f is not None else pm.div(torch.norm(pm, p=2, dim=2).unsqueeze(2))
            features = ChannelMajorOf.wrap(pm)
        else:
            features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))  # SpaCapNet.py:66-67 (no eps)
        data_dict["vote_xyz"] = xyz
        data_dict["vote_features"] = features
        data_dict = self.proposal(xyz, features, data_dict)
        if after_proposal is not None:
            after_proposal(data_dict)
        if not self.no_caption:
            data_dict = self.caption(data_dict, is_eval)
        return data_dict


def build_default(vocab_size=3001, input_feature_dim=1, num_proposal=256, N=6, h=8, d_model=128, d_ff=2048,
                  dropout=0.1, mean_size_arr=None):
    """The configuration scripts/train.py builds with default flags (xyz + height, encoder on, early guide,
    relation head, learned xyz position encoding; scripts/train.py:125-156)."""
    from . import synthetic as S
    msa = mean_size_arr if mean_size_arr is not None else S.mean_size_arr().numpy()
    return SpaCapNet(num_class=S.NUM_CLASS, vocabulary=S.make_vocabulary(vocab_size),
                     num_heading_bin=S.NUM_HEADING_BIN, num_size_cluster=S.NUM_SIZE_CLUSTER, mean_size_arr=msa,
                     input_feature_dim=input_feature_dim, num_proposal=num_proposal, N=N, h=h, d_model=d_model,
                     d_ff=d_ff, transformer_dropout=dropout, src_pos_type="xyz", use_transformer_encoder=True,
                     early_guide=True, check_relation=True)

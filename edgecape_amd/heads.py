"""Config-holding counterparts of the reference's head modules.

In the reference these are nn.Modules whose forward does the arithmetic.  Here they only validate and
carry hyper-parameters (same constructor signatures, same registry names, same asserts) — the arithmetic
lives in libedgecape_hip.so and is driven by `EdgeCape.forward` (detector.py).
"""
from .registry import HEADS, POSITIONAL_ENCODING, TRANSFORMER, build_head, build_positional_encoding, build_transformer


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding:
    """positional_encoding.py:11-55."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * 3.141592653589793, eps=1e-6, offset=0.,
                 init_cfg=None):
        if normalize:
            assert isinstance(scale, (float, int)), "when normalize is set, scale should be provided and in float or int type"
        self.num_feats, self.temperature, self.normalize, self.scale, self.eps, self.offset = \
            num_feats, temperature, normalize, scale, eps, offset
        # the HIP kernels implement exactly the shipped configuration
        if temperature != 10000 or not normalize or offset != 0. or abs(scale - 2 * 3.141592653589793) > 1e-9:
            raise NotImplementedError("only SinePositionalEncoding(temperature=1e4, normalize=True, scale=2*pi) is built")


@TRANSFORMER.register_module()
class TwoStageSupportRefineTransformer:
    """encoder_decoder.py:115-175."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, similarity_proj_dim=256, dynamic_proj_dim=128,
                 return_intermediate_dec=True, attn_bias=False, max_hops=5, use_bias_attn_module=False,
                 masked_supervision=False, recon_features=False):
        self.d_model, self.nhead = d_model, nhead
        self.num_encoder_layers, self.num_decoder_layers = num_encoder_layers, num_decoder_layers
        self.dim_feedforward, self.dropout, self.activation = dim_feedforward, dropout, activation
        self.similarity_proj_dim, self.dynamic_proj_dim = similarity_proj_dim, dynamic_proj_dim
        self.attn_bias, self.max_hops, self.use_bias_attn_module = attn_bias, max_hops, use_bias_attn_module
        self.masked_supervision = masked_supervision
        if activation != "relu":
            raise RuntimeError(f"activation should be relu/gelu, not {activation}." if activation not in ("gelu", "glu")
                               else "only activation='relu' is built")
        unsupported = []
        if d_model != 256 or nhead != 8: unsupported.append("d_model/nhead != 256/8")
        if normalize_before: unsupported.append("normalize_before=True")
        if not return_intermediate_dec: unsupported.append("return_intermediate_dec=False")
        # attn_bias=False (stage-1 / stage-2 models of run.py:44-88): the decoder layers' self-attention adds no Markov bias - plain
        # nn.MultiheadAttention, or BiasedMultiheadAttention(bias_attn=False) with use_bias_attn_module (encoder_decoder.py:551-560)
        if attn_bias and max_hops != 4: unsupported.append("max_hops != 4")
        if similarity_proj_dim != 256 or dynamic_proj_dim > 128: unsupported.append("proposal generator dims")
        if unsupported:
            raise NotImplementedError("HIP path implements the shipped test configs only: " + ", ".join(unsupported))


@HEADS.register_module()
class SkeletonPredictor:
    """skeleton.py:9-56."""

    def __init__(self, d_model=256, nhead=8, num_layers=3, dim_feedforward=384, dropout=0.1, activation="relu",
                 normalize_before=False, learn_skeleton=False, max_hop=5, adj_normalization=True, markov_bias=True,
                 mask_res=False, use_zero_conv=True, max_hops=4, two_way_attn=True, gcn_norm=False):
        self.d_model, self.nhead, self.num_layers, self.dim_feedforward = d_model, nhead, num_layers, dim_feedforward
        self.learn_skeleton, self.max_hop = learn_skeleton, max_hop
        # learn_skeleton=False (skeleton.py:70-74): the normalised ground-truth adjacency, none of the layers below runs
        if learn_skeleton and (not (adj_normalization and use_zero_conv and two_way_attn) or mask_res or gcn_norm
                               or activation != "relu" or normalize_before):
            raise NotImplementedError("HIP path implements SkeletonPredictor(learn_skeleton=True) with default flags only")

    def init_weights(self):
        pass


@HEADS.register_module()
class TwoStageHead:
    """head.py:61-141."""

    def __init__(self, in_channels, transformer=None, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                                                              normalize=True),
                 share_kpt_branch=False, num_decoder_layer=3, with_heatmap_loss=False, heatmap_loss_weight=2.0,
                 skeleton_loss_weight=1, train_cfg=None, test_cfg=None, skeleton_head=None, learn_skeleton=False,
                 masked_supervision=False, freeze=None, model_freeze=None, masking_ratio=0.5):
        self.in_channels = in_channels
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.d_model
        assert "num_feats" in positional_encoding
        num_feats = positional_encoding["num_feats"]
        assert num_feats * 2 == self.embed_dims, "embed_dims should" \
            f" be exactly 2 times of num_feats. Found {self.embed_dims}" f" and {num_feats}."
        self.share_kpt_branch, self.num_decoder_layer = share_kpt_branch, num_decoder_layer
        self.train_cfg = {} if train_cfg is None else train_cfg
        self.test_cfg = {} if test_cfg is None else test_cfg
        self.target_type = self.test_cfg.get("target_type", "GaussianHeatMap")
        skeleton_head = dict(skeleton_head)
        skeleton_head["max_hop"] = transformer.get("max_hops", 4)   # head.py:122
        self.skeleton_head = build_head(skeleton_head)
        self.learn_skeleton = learn_skeleton
        if num_decoder_layer != self.transformer.num_decoder_layers:
            raise NotImplementedError("num_decoder_layer must equal transformer.num_decoder_layers")

    def init_weights(self):
        pass

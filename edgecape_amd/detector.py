"""`EdgeCape` — the reference's detector API (EdgeCape/models/detectors/EdgeCape.py) over the HIP library.

Same registry name, constructor keys and `forward(..., return_loss=False)` result dict, so
`configs/test/*.py` build unchanged and the reference's callers (`single_gpu_test`,
gradio `process`, demo) can swap it in.  No torch.nn.Module compute: `forward_test` hands raw device
pointers to `libedgecape_hip.so` (engine.py) and does only the reference's host-side `decode`.
"""
import numpy as np
import torch

from . import _lib
from .engine import HipEngine, normalize_state_dict
from .registry import POSENETS, build_head
from .synth import ARCHS
from . import heads  # noqa: F401  (registers TwoStageHead & co.)


@POSENETS.register_module()
class EdgeCape:
    def __init__(self, keypoint_head, encoder_config=None, train_cfg=None, test_cfg=None, pretrained="dinov2_vits14",
                 backbone_precision="fp32", head_precision="fp32", max_batch=None):
        if pretrained not in ARCHS:
            raise KeyError(f"unknown backbone {pretrained!r}; expected one of {sorted(ARCHS)}")
        self.pretrained = pretrained
        self.backbone = "dinov2"
        self.keypoint_head_module = build_head(keypoint_head)   # validates the config like the reference's build_head
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg if test_cfg is not None else {}
        self.target_type = self.test_cfg.get("target_type", "GaussianHeatMap")
        if self.keypoint_head_module.in_channels != ARCHS[pretrained]["C"]:
            raise ValueError(f"keypoint_head.in_channels={self.keypoint_head_module.in_channels} does not match the "
                             f"{pretrained} width {ARCHS[pretrained]['C']}")
        self.backbone_precision, self.head_precision = backbone_precision, head_precision
        self._max_batch = max_batch
        self._state_dict = None
        self._engines = {}
        self.training = False

    # ---- nn.Module-like surface used by the reference's callers -------------------------------------
    @property
    def with_keypoint(self):
        return hasattr(self, "keypoint_head_module")

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope of the MI355X inference path (SURVEY §8)")
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def init_weights(self):
        pass

    def load_state_dict(self, state_dict, strict=True):
        sd = normalize_state_dict(state_dict)
        if strict:
            need = ["encoder_query.pos_embed", "encoder_query.patch_embed.proj.weight", "keypoint_head_module.input_proj.weight"]
            missing = [k for k in need if k not in sd]
            if missing:
                raise KeyError(f"missing keys in state dict: {missing}")
        self._state_dict = sd
        self._engines.clear()
        return self

    def state_dict(self):
        return self._state_dict

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # ---- engine cache --------------------------------------------------------------------------------
    def _engine(self, image_size, bs, shots, K):
        if self._state_dict is None:
            raise RuntimeError("no weights loaded: call load_state_dict() / load_checkpoint() first")
        for key, eng in self._engines.items():
            if key[0] == image_size and key[3] == K and key[1] >= bs and key[2] >= shots:
                return eng
        mb = max(bs, self._max_batch or 0)
        th = self.keypoint_head_module.transformer
        eng = HipEngine(self._state_dict, arch=self.pretrained, image_size=image_size, max_batch=mb, max_shots=shots,
                        num_kpts=K, ffn_dim=th.dim_feedforward,
                        skel_ffn_dim=self.keypoint_head_module.skeleton_head.dim_feedforward,
                        backbone_precision=self.backbone_precision, head_precision=self.head_precision,
                        enc_layers=th.num_encoder_layers, dec_layers=th.num_decoder_layers,
                        skel_layers=self.keypoint_head_module.skeleton_head.num_layers, max_hops=th.max_hops)
        self._engines[(image_size, mb, shots, K)] = eng
        return eng

    # ---- reference API ----------------------------------------------------------------------------
    def forward(self, img_s, img_q, target_s=None, target_weight_s=None, target_q=None, target_weight_q=None,
                img_metas=None, return_loss=True, **kwargs):
        """EdgeCape.forward (EdgeCape.py:56-80)."""
        if return_loss:
            raise NotImplementedError("forward_train is out of scope of the MI355X inference path (SURVEY §8)")
        return self.forward_test(img_s, target_s, target_weight_s, img_q, target_q, target_weight_q, img_metas, **kwargs)

    def predict(self, img_s, target_s, target_weight_s, img_q, img_metas=None):
        """EdgeCape.predict (EdgeCape.py:165-184); returns device tensors."""
        bs, _, H, W = img_q.shape
        if H != W:
            raise ValueError("square inputs only")
        K = target_s[0].shape[1]
        mask_s = torch.as_tensor(target_weight_s[0]).float()
        for tw in target_weight_s:                    # EdgeCape.py:175-177
            mask_s = mask_s * torch.as_tensor(tw).float()
        skeletons = [m["sample_skeleton"][0] for m in img_metas]   # EdgeCape.py:179
        eng = self._engine(H, bs, len(img_s), K)
        o = eng.forward(img_q, img_s, target_s, mask_s, skeletons)
        return o["output_kpts"], o["initial_proposals"], o["similarity_map"], mask_s, None, o["adj"]

    def forward_test(self, img_s, target_s, target_weight_s, img_q, target_q=None, target_weight_q=None, img_metas=None,
                     vis_offset=True, **kwargs):
        """EdgeCape.forward_test (EdgeCape.py:131-163)."""
        batch_size, _, img_height, img_width = img_q.shape
        output, initial_proposals, similarity_map, mask_s, _, adj = self.predict(img_s, target_s, target_weight_s, img_q, img_metas)
        predicted_pose = output[-1].cpu().numpy()             # device -> host boundary (EdgeCape.py:150)
        result = {}
        result.update(self.decode(img_metas, predicted_pose, img_size=[img_width, img_height]))
        if vis_offset:
            result.update({"points": torch.cat((initial_proposals[None], output)).cpu().numpy()})
        result.update({"sample_image_file": [img_metas[i]["sample_image_file"] for i in range(len(img_metas))]})
        result.update({"skeleton": adj[0].cpu().numpy()})
        return result

    def decode(self, img_metas, output, img_size, **kwargs):
        """TwoStageHead.decode + transform_preds (head.py:324-387, post_transforms.py:150-194), host numpy."""
        batch_size = len(img_metas)
        W, H = img_size
        output = output * np.array([W, H])[None, None, :]
        bbox_ids = []
        c = np.zeros((batch_size, 2), dtype=np.float32)
        s = np.zeros((batch_size, 2), dtype=np.float32)
        image_paths = []
        score = np.ones(batch_size)
        for i in range(batch_size):
            c[i, :] = img_metas[i]["query_center"]
            s[i, :] = img_metas[i]["query_scale"]
            image_paths.append(img_metas[i]["query_image_file"])
            if "query_bbox_score" in img_metas[i]:
                score[i] = float(np.array(img_metas[i]["query_bbox_score"]).reshape(-1)[0])
            if "bbox_id" in img_metas[i]:
                bbox_ids.append(img_metas[i]["bbox_id"])
            elif "query_bbox_id" in img_metas[i]:
                bbox_ids.append(img_metas[i]["query_bbox_id"])
        preds = np.zeros(output.shape)
        use_udp = self.test_cfg.get("use_udp", False)
        for idx in range(output.shape[0]):
            scale = s[idx] * 200.0
            if use_udp:
                sx, sy = scale[0] / (W - 1.0), scale[1] / (H - 1.0)
            else:
                sx, sy = scale[0] / W, scale[1] / H
            preds[idx, :, 0] = output[idx, :, 0] * sx + c[idx, 0] - scale[0] * 0.5
            preds[idx, :, 1] = output[idx, :, 1] * sy + c[idx, 1] - scale[1] * 0.5
        all_preds = np.zeros((batch_size, preds.shape[1], 3), dtype=np.float32)
        all_boxes = np.zeros((batch_size, 6), dtype=np.float32)
        all_preds[:, :, 0:2] = preds[:, :, 0:2]
        all_preds[:, :, 2:3] = 1.0
        all_boxes[:, 0:2] = c[:, 0:2]
        all_boxes[:, 2:4] = s[:, 0:2]
        all_boxes[:, 4] = np.prod(s * 200.0, axis=1)
        all_boxes[:, 5] = score
        return dict(preds=all_preds, boxes=all_boxes, image_paths=image_paths, bbox_ids=bbox_ids)


from .checkpoint import load_checkpoint  # noqa: E402,F401  (mmcv.runner.load_checkpoint stand-in, test.py:124)


def hip_library_loaded():
    """True when libedgecape_hip.so is mapped into this process (used by tests to prove the native path ran)."""
    with open("/proc/self/maps") as f:
        return "libedgecape_hip.so" in f.read()


__all__ = ["EdgeCape", "load_checkpoint", "hip_library_loaded", "_lib"]

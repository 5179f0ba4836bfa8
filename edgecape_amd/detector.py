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
        self._engines = {}          # (image_size, max_batch, shots, K) -> HipEngine, least recently used first
        self.max_engines = 4
        self.training = False
        self._episode_slots = 0     # > 0: enable_episode_cache() - the support side is computed once per support set
        self._episodes = {}         # id(engine) -> dict(cache=SupportCache, slot_of=OrderedDict(support key -> slot))

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
        self._episodes.clear()
        return self

    def state_dict(self):
        return self._state_dict

    # ---- support-side episode cache through the reference API --------------------------------------
    def enable_episode_cache(self, max_episodes=64):
        """The reference's evaluation pairs ONE support set with 15 consecutive queries (test_dataset.py:86-99) and its forward_test
        recomputes the support backbone features and the whole skeleton head for every pair.  With the cache on, forward_test / submit
        recognise a support set they have seen (by its annotations' image files, crop boxes and keypoints in img_metas) and only encode
        the new ones of a batch - in the SAME backbone pass as the batch's queries (ec_forward_episodes).  Results are those of the
        plain path; `max_episodes` support sets are kept (least recently used out).  0 switches it off."""
        self._episode_slots = int(max_episodes)
        self._episodes.clear()
        return self

    @staticmethod
    def _support_key(meta):
        """Identity of a pair's support set: per shot the annotation's image file plus every annotation field img_metas carries - crop box
        (centre, scale, rotation), keypoints and their visibility (which determine target_s / mask_s), bbox id.  None when the metas hold
        NOTHING beyond the file names (a custom Collect, demo.py's `sample_image_file=['']`): two different annotations of one image
        would then share a key and the second would silently get the first one's cached tokens - such a batch takes the plain path."""
        def raw(v):
            if isinstance(v, torch.Tensor):                        # (demo.py hands sample_joints_3d over as a CUDA tensor)
                v = v.detach().cpu().numpy()
            return np.ascontiguousarray(np.asarray(v)).tobytes()
        parts, discriminating = [], False
        for s, f in enumerate(meta["sample_image_file"]):
            p = [str(f)]
            for k in ("sample_center", "sample_scale", "sample_rotation", "sample_joints_3d", "sample_joints_3d_visible", "sample_bbox_id"):
                if k in meta:
                    p.append(k)
                    p.append(raw(meta[k][s]))
                    discriminating = discriminating or k in ("sample_center", "sample_scale", "sample_joints_3d", "sample_bbox_id")
            parts.append(tuple(p))
        if not discriminating:
            return None
        return (tuple(parts), raw(np.asarray(meta["sample_skeleton"][0], np.int64)))

    def _device_forward(self, eng, img_q, img_s, target_s, mask_s, img_metas, pipelined=False):
        """Device part of one batch: plain (ec_forward / ec_forward_pipelined), or through the episode cache."""
        bs, K = img_q.shape[0], target_s[0].shape[1]
        skeletons = [m["sample_skeleton"][0] for m in img_metas]   # EdgeCape.py:179
        keys = [self._support_key(m) for m in img_metas] if self._episode_slots > 0 else None
        if keys is None or any(k is None for k in keys) or len(set(keys)) > self._episode_slots:
            if not pipelined:
                return eng.forward(img_q, img_s, target_s, mask_s, skeletons)
            iq, is_, ts = eng._dev(img_q), [eng._dev(x) for x in img_s], [eng._dev(t) for t in target_s]
            ms = eng._dev(mask_s).reshape(bs, K)
            edges, off = eng._edges(skeletons, bs)
            o = eng.forward_pipelined(iq, is_, ts, ms, edges, off, eng._outputs(bs))
            o["_keep"] = (iq, is_, ts, ms)
            return o
        from collections import OrderedDict
        st = self._episodes.get(id(eng))
        if st is None:
            st = self._episodes[id(eng)] = dict(cache=eng.support_cache(self._episode_slots), slot_of=OrderedDict())
        slot_of = st["slot_of"]
        first = {}
        for i, k in enumerate(keys):
            first.setdefault(k, i)
        new_keys = [k for k in first if k not in slot_of]
        used = set(slot_of.values())
        free = [s for s in range(self._episode_slots) if s not in used]
        for k in list(slot_of):                                   # least recently used first; never a support set of this batch
            if len(free) >= len(new_keys):
                break
            if k not in first:
                free.append(slot_of.pop(k))
        new = None
        if new_keys:
            idx = [first[k] for k in new_keys]
            for k in new_keys:
                slot_of[k] = free.pop(0)
            take = (lambda x: x[idx]) if not isinstance(img_s[0], torch.Tensor) else (lambda x: x[torch.as_tensor(idx, device=x.device)])
            mask_np = mask_s[torch.as_tensor(idx)] if isinstance(mask_s, torch.Tensor) else np.asarray(mask_s)[idx]
            new = dict(img_s=[take(x) for x in img_s], target_s=[take(x) for x in target_s], mask_s=mask_np,
                       skeletons=[skeletons[i] for i in idx], slots=[slot_of[k] for k in new_keys])
        for k in first:
            slot_of.move_to_end(k)
        return eng.forward_episodes(st["cache"], img_q, [slot_of[k] for k in keys], new=new, pipelined=pipelined)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # ---- engine cache --------------------------------------------------------------------------------
    def _engine(self, image_size, bs, shots, K):
        if self._state_dict is None:
            raise RuntimeError("no weights loaded: call load_state_dict() / load_checkpoint() first")
        for key, eng in self._engines.items():
            if key[0] == image_size and key[3] == K and key[1] >= bs and key[2] >= shots:
                self._engines[key] = self._engines.pop(key)      # most recently used goes last
                return eng
        # every engine owns a private copy of the weights and its workspace on the GPU: keep at most `max_engines` of them
        # (K is dynamic in the demo path - one engine per clicked-point count would otherwise grow without bound)
        while len(self._engines) >= self.max_engines:
            old = self._engines.pop(next(iter(self._engines)))    # least recently used; its __del__ frees the device memory
            self._episodes.pop(id(old), None)
        mb = max(bs, self._max_batch or 0)
        th = self.keypoint_head_module.transformer
        eng = HipEngine(self._state_dict, arch=self.pretrained, image_size=image_size, max_batch=mb, max_shots=shots,
                        num_kpts=K, ffn_dim=th.dim_feedforward,
                        skel_ffn_dim=self.keypoint_head_module.skeleton_head.dim_feedforward,
                        backbone_precision=self.backbone_precision, head_precision=self.head_precision,
                        enc_layers=th.num_encoder_layers, dec_layers=th.num_decoder_layers,
                        skel_layers=self.keypoint_head_module.skeleton_head.num_layers, max_hops=th.max_hops if th.attn_bias else 4,
                        learn_skeleton=self.keypoint_head_module.skeleton_head.learn_skeleton, attn_bias=th.attn_bias)
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
        bs, _, H, W = img_q.shape                     # any height / width (EdgeCape.py:143); an engine per input size
        K = target_s[0].shape[1]
        mask_s = torch.as_tensor(target_weight_s[0]).float()
        for tw in target_weight_s:                    # EdgeCape.py:175-177
            mask_s = mask_s * torch.as_tensor(tw).float()
        eng = self._engine(H if H == W else (H, W), bs, len(img_s), K)
        o = self._device_forward(eng, img_q, img_s, target_s, mask_s, img_metas)
        return o["output_kpts"], o["initial_proposals"], o["similarity_map"], mask_s, None, o["adj"]

    def forward_test(self, img_s, target_s, target_weight_s, img_q, target_q=None, target_weight_q=None, img_metas=None,
                     vis_offset=True, **kwargs):
        """EdgeCape.forward_test (EdgeCape.py:131-163): same result dict (preds / boxes / image_paths / bbox_ids, points,
        sample_image_file, skeleton).  The three device results the dict needs leave the GPU as three asynchronous copies
        behind ONE stream synchronisation (the reference synchronises three times through .cpu())."""
        height, width = img_q.shape[-2:]
        output, initial_proposals, similarity_map, mask_s, _, adj = self.predict(img_s, target_s, target_weight_s, img_q, img_metas)
        host = [t.to("cpu", non_blocking=True) for t in (output, initial_proposals, adj[0])]
        torch.cuda.current_stream().synchronize()                 # device -> host boundary (EdgeCape.py:150)
        layers, proposals, skeleton = (h.numpy() for h in host)
        result = self.decode(img_metas, layers[-1], img_size=[width, height])
        if vis_offset:
            result["points"] = np.concatenate((proposals[None], layers), 0)
        result["sample_image_file"] = [m["sample_image_file"] for m in img_metas]
        result["skeleton"] = skeleton
        return result

    # ---- pipelined evaluation (ec_forward_pipelined): submit batch i+1 before collecting batch i --------------------
    def submit(self, img_s, target_s, target_weight_s, img_q, target_q=None, target_weight_q=None, img_metas=None, vis_offset=True,
               **kwargs):
        """First half of forward_test for a back-to-back loop (apis.single_gpu_test(pipelined=True)): enqueue the device work of one
        batch through ec_forward_pipelined and the device -> host copies of its results on a copy stream that waits for THIS
        batch's head only; returns a ticket for collect().  The batch's head then overlaps the next submit()'s backbone."""
        height, width = img_q.shape[-2:]
        bs, K = img_q.shape[0], target_s[0].shape[1]
        mask_s = torch.as_tensor(target_weight_s[0]).float()
        for tw in target_weight_s:
            mask_s = mask_s * torch.as_tensor(tw).float()
        eng = self._engine(height if height == width else (height, width), bs, len(img_s), K)
        o = self._device_forward(eng, img_q, img_s, target_s, mask_s, img_metas, pipelined=True)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream()
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream())      # this call's backbone ...
        eng.pipeline_flush(cs)                           # ... and its head, not the next call's backbone
        # PINNED destinations: an asynchronous device -> host copy into pageable memory blocks the calling thread until the copy has
        # run - i.e. until this batch's head has finished - and submit(i + 1) could then never enqueue backbone i + 1 beside head i.
        # The buffers come from a small free list and go back to it in collect().
        src = (o["output_kpts"], o["initial_proposals"], o["adj"][0])
        host = [self._pinned(t.shape) for t in src]
        with torch.cuda.stream(cs):
            for h, t in zip(host, src):
                h.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cs)
        return dict(host=host, done=done, keep=o, img_metas=img_metas, size=[width, height], vis_offset=vis_offset)

    def _pinned(self, shape):
        """A pinned float32 host tensor of `shape` from the free list (page-locking is a system call: the loop reuses its buffers)."""
        pool = self.__dict__.setdefault("_pin_pool", {})
        free = pool.setdefault(tuple(shape), [])
        return free.pop() if free else torch.empty(tuple(shape), dtype=torch.float32, pin_memory=True)

    def collect(self, ticket):
        """Second half: wait for the ticket's copies and build the reference's result dict (forward_test's host part)."""
        ticket["done"].synchronize()
        layers, proposals, skeleton = (h.numpy().copy() for h in ticket["host"])    # own copies: the pinned buffers are reused
        for h in ticket["host"]:
            self._pin_pool[tuple(h.shape)].append(h)
        ticket["host"] = None
        img_metas = ticket["img_metas"]
        result = self.decode(img_metas, layers[-1], img_size=ticket["size"])
        if ticket["vis_offset"]:
            result["points"] = np.concatenate((proposals[None], layers), 0)
        result["sample_image_file"] = [m["sample_image_file"] for m in img_metas]
        result["skeleton"] = skeleton
        ticket["keep"] = None
        return result

    def decode(self, img_metas, output, img_size, **kwargs):
        """TwoStageHead.decode + transform_preds (head.py:324-387, post_transforms.py:150-194) for the whole batch at once:
        normalised [bs,K,2] coordinates -> pixels of the model input -> pixels of the source image through each query's
        (center, scale) box.  dtypes follow the reference (float32 box, float64 coordinates) so `boxes` is bit-identical."""
        n = len(img_metas)
        size = np.asarray(img_size, np.float64)                                         # [W, H]
        center = np.array([m["query_center"] for m in img_metas], np.float32).reshape(n, 2)
        scale = np.array([m["query_scale"] for m in img_metas], np.float32).reshape(n, 2)
        score = np.array([float(np.asarray(m["query_bbox_score"]).reshape(-1)[0]) if "query_bbox_score" in m else 1.0
                          for m in img_metas], np.float32)
        box = scale * np.float32(200.0)                                                 # bbox side lengths in source pixels
        denom = (size - 1.0) if self.test_cfg.get("use_udp", False) else size
        per_px = box / denom.astype(np.float32)                                         # source pixels per model-input pixel
        xy = np.asarray(output, np.float64) * size                                      # model-input pixels
        xy = xy * per_px[:, None, :] + center[:, None, :] - (box * np.float32(0.5))[:, None, :]
        preds = np.ones((n, xy.shape[1], 3), np.float32)
        preds[:, :, :2] = xy
        boxes = np.concatenate([center, scale, np.prod(box, axis=1, keepdims=True), score[:, None]], 1).astype(np.float32)
        ids = [m["bbox_id"] if "bbox_id" in m else m["query_bbox_id"] for m in img_metas if "bbox_id" in m or "query_bbox_id" in m]
        return dict(preds=preds, boxes=boxes, image_paths=[m["query_image_file"] for m in img_metas], bbox_ids=ids)


from .checkpoint import load_checkpoint  # noqa: E402,F401  (mmcv.runner.load_checkpoint stand-in, test.py:124)


def hip_library_loaded():
    """True when libedgecape_hip.so is mapped into this process (used by tests to prove the native path ran)."""
    with open("/proc/self/maps") as f:
        return "libedgecape_hip.so" in f.read()


__all__ = ["EdgeCape", "load_checkpoint", "hip_library_loaded", "_lib"]

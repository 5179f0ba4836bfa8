"""HipEngine: one ec_handle (include/edgecape_hip.h) = weights + workspace for one
(backbone arch, image size, max batch, max shots, precision).  Thin plumbing around the C ABI:
torch is used only to own device buffers and the stream; no torch op computes anything here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .posembed import interpolate_pos_embed
from .synth import ARCHS, PATCH

_TORCH_DT = {torch.float32: _lib.EC_DT_F32, torch.float16: _lib.EC_DT_F16, torch.bfloat16: _lib.EC_DT_BF16,
             torch.float64: _lib.EC_DT_F64}
_NP_DT = {np.dtype("float32"): _lib.EC_DT_F32, np.dtype("float16"): _lib.EC_DT_F16, np.dtype("float64"): _lib.EC_DT_F64}


def _host_array(v):
    """state-dict value (numpy / torch, any float dtype) -> (contiguous host buffer keeper, ptr, shape, dtype code)."""
    if isinstance(v, torch.Tensor):
        t = v.detach().cpu().contiguous()
        if t.dtype not in _TORCH_DT:
            t = t.float()
        return t, t.data_ptr(), tuple(t.shape), _TORCH_DT[t.dtype]
    a = np.ascontiguousarray(v)
    if a.dtype not in _NP_DT:
        a = a.astype(np.float32)
    return a, a.ctypes.data, tuple(a.shape), _NP_DT[a.dtype]


def normalize_state_dict(sd):
    """Reference checkpoint key handling (test.py:124, EdgeCape.py:36, bias_attn.py:236-265):
    strip an outer 'state_dict', map encoder_sample.* onto encoder_query.* (same module twice), and split a
    fused decoder self_attn.in_proj_{weight,bias} (stage-2 checkpoints) into q/k/v_proj."""
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        if k.startswith("encoder_sample."):
            k2 = "encoder_query." + k[len("encoder_sample."):]
            if k2 in sd:
                continue
            k = k2
        out[k] = v
    for k in list(out.keys()):
        if ".transformer.decoder.layers." in k and k.endswith("self_attn.in_proj_weight"):
            p = k[: -len("in_proj_weight")]
            w = out.pop(k)
            w = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
            dd = w.shape[0] // 3
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[p + n + ".weight"] = w[i * dd:(i + 1) * dd]
            if p + "in_proj_bias" in out:
                b = out.pop(p + "in_proj_bias")
                b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[p + n + ".bias"] = b[i * dd:(i + 1) * dd]
    return out


class SupportCache:
    """Owner of an ec_support_t (cached support-side state of up to `max_episodes` episodes of one engine)."""

    def __init__(self, engine, max_episodes):
        self.engine, self.n_episodes = engine, 0
        h = C.c_void_p()
        _lib.check(engine.lib.ec_support_create(engine.h, max_episodes, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.engine.lib.ec_support_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HipEngine:
    def __init__(self, state_dict, arch="dinov2_vits14", image_size=224, max_batch=2, max_shots=1, num_kpts=100,
                 ffn_dim=384, skel_ffn_dim=None, backbone_precision="fp32", head_precision="fp32", heatmap_size=64,
                 d_model=256, nhead=8, enc_layers=3, dec_layers=3, skel_layers=3, max_hops=4, learn_skeleton=True, attn_bias=True):
        if not torch.cuda.is_available():
            raise _lib.EdgeCapeHipError("no MI355X / HIP device visible: the EdgeCape hot path has no CPU fallback")
        self.lib = _lib.load()
        a = ARCHS[arch]
        # image_size: an int (square) or (H, W) - the reference takes any img.shape[-2:] (EdgeCape.py:143)
        H, Wd = (image_size, image_size) if isinstance(image_size, (int, np.integer)) else (int(image_size[0]), int(image_size[1]))
        self.arch, self.C, self.image_size = arch, a["C"], (H if H == Wd else (H, Wd))
        self.gh, self.gw = H // PATCH, Wd // PATCH
        self.g = self.gh if self.gh == self.gw else (self.gh, self.gw)
        self.HW = self.gh * self.gw
        self.K, self.max_batch, self.max_shots = num_kpts, max_batch, max_shots
        self.dec_layers, self.max_hops = dec_layers, max_hops
        prec = {"fp32": _lib.EC_F32, "bf16": _lib.EC_BF16, "bf16x3": _lib.EC_BF16X3, "fp16": _lib.EC_F16, "mixed": _lib.EC_MIXED, "fp16x2": _lib.EC_F16X2}
        if backbone_precision not in ("fp32", "bf16x3", "fp16x2", "bf16", "fp16") or head_precision not in ("fp32", "bf16x3", "mixed"):
            raise ValueError("backbone_precision must be fp32 / bf16x3 / fp16x2 / bf16 / fp16 and head_precision fp32 / bf16x3 / mixed")
        cfg = _lib.EcConfig(embed_dim=a["C"], depth=a["depth"], num_heads=a["heads"], image_size=H, image_width=(0 if H == Wd else Wd), patch=PATCH,
                            num_kpts=num_kpts, d_model=d_model, nhead=nhead, enc_layers=enc_layers, dec_layers=dec_layers,
                            skel_layers=skel_layers, ffn_dim=ffn_dim, skel_ffn_dim=skel_ffn_dim or a["C"], max_hops=max_hops,
                            heatmap_size=heatmap_size, max_shots=max_shots, max_batch=max_batch,
                            backbone_precision=prec[backbone_precision], head_precision=prec[head_precision],
                            gt_skeleton=0 if learn_skeleton else 1, no_attn_bias=0 if attn_bias else 1)
        self.learn_skeleton, self.attn_bias = bool(learn_skeleton), bool(attn_bias)
        self.backbone_precision, self.head_precision = backbone_precision, head_precision
        h = C.c_void_p()
        _lib.check(self.lib.ec_create(C.byref(cfg), C.byref(h)))
        self.h = h
        sd = normalize_state_dict(state_dict)
        pos = None
        for name, v in sd.items():
            if not (name.startswith("encoder_query.") or name.startswith("keypoint_head_module.")):
                continue
            if name == "encoder_query.pos_embed":
                pv = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
                pos = interpolate_pos_embed(pv, (self.gh, self.gw))
                continue
            keep, ptr, shape, dt = _host_array(v)
            shp = (C.c_int64 * len(shape))(*shape)
            _lib.check(self.lib.ec_load_tensor(self.h, name.encode(), C.c_void_p(ptr), shp, len(shape), dt))
        if pos is None:
            raise KeyError("state dict has no encoder_query.pos_embed")
        pos = np.ascontiguousarray(pos, np.float32)
        _lib.check(self.lib.ec_set_pos_embed(self.h, C.c_void_p(pos.ctypes.data), pos.shape[0], pos.shape[1]))
        _lib.check(self.lib.ec_finalize(self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- helpers ---------------------------------------------------------------------------------
    @staticmethod
    def _dev(x):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        return t.to(device="cuda", dtype=torch.float32).contiguous()

    @staticmethod
    def _edges(skeletons, bs):
        flat, off = [], [0]
        for b in range(bs):
            e = np.asarray(skeletons[b], np.int64)
            if e.ndim == 2 and e.shape[0] > 0:   # skeleton.py:177 — an empty list contributes no edges
                flat.append(e.reshape(-1, 2))
                off.append(off[-1] + e.shape[0])
            else:
                off.append(off[-1])
        edges = np.ascontiguousarray(np.concatenate(flat, 0) if flat else np.zeros((0, 2)), np.int32)
        return edges, np.asarray(off, np.int32)

    def _outputs(self, bs):
        dev = "cuda"
        o = dict(output_kpts=torch.empty(self.dec_layers, bs, self.K, 2, device=dev),
                 initial_proposals=torch.empty(bs, self.K, 2, device=dev),
                 similarity_map=torch.empty(bs, self.K, self.gh, self.gw, device=dev),
                 adj=torch.empty(bs, 2, self.K, self.K, device=dev),
                 attn_adj=torch.empty(self.max_hops + 1, bs, self.K, self.K, device=dev),
                 out_points=torch.empty(self.dec_layers + 1, bs, self.K, 2, device=dev))
        eo = _lib.EcOutputs(o["output_kpts"].data_ptr(), o["initial_proposals"].data_ptr(), o["similarity_map"].data_ptr(),
                            o["adj"].data_ptr(), o["attn_adj"].data_ptr(), o["out_points"].data_ptr())
        return o, eo

    @staticmethod
    def _ptr_array(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    # ---- entry points ----------------------------------------------------------------------------
    def backbone(self, img, nchw=True):
        """EdgeCape.extract_features for one image batch -> [n,C,g,g] (or [n,HW,C] tokens)."""
        img = self._dev(img)
        n = img.shape[0]
        out = torch.empty((n, self.C, self.gh, self.gw) if nchw else (n, self.HW, self.C), device="cuda")
        _lib.check(self.lib.ec_backbone(self.h, img.data_ptr(), n, out.data_ptr(),
                                        _lib.EC_LAYOUT_NCHW if nchw else _lib.EC_LAYOUT_TOKENS, _lib.current_stream()))
        return out

    def head(self, feature_q, feature_s, target_s, mask_s, skeletons):
        """TwoStageHead.forward on NCHW features (the reference's layout)."""
        fq = self._dev(feature_q)
        fs = [self._dev(f) for f in feature_s]
        ts = [self._dev(t) for t in target_s]
        ms = self._dev(mask_s).reshape(fq.shape[0], self.K)
        bs, S = fq.shape[0], len(fs)
        edges, off = self._edges(skeletons, bs)
        o, eo = self._outputs(bs)
        _lib.check(self.lib.ec_head(self.h, fq.data_ptr(), self._ptr_array(fs), _lib.EC_LAYOUT_NCHW, self._ptr_array(ts),
                                    ms.data_ptr(), edges.ctypes.data, off.ctypes.data, bs, S, _lib.current_stream(),
                                    C.byref(eo)))
        return o

    def forward(self, img_q, img_s, target_s, mask_s, skeletons):
        """Device part of EdgeCape.predict: backbone on query + supports, then the head. Asynchronous."""
        iq = self._dev(img_q)
        is_ = [self._dev(x) for x in img_s]
        ts = [self._dev(t) for t in target_s]
        bs, S = iq.shape[0], len(is_)
        ms = self._dev(mask_s).reshape(bs, self.K)
        edges, off = self._edges(skeletons, bs)
        o, eo = self._outputs(bs)
        _lib.check(self.lib.ec_forward(self.h, iq.data_ptr(), self._ptr_array(is_), self._ptr_array(ts), ms.data_ptr(),
                                       edges.ctypes.data, off.ctypes.data, bs, S, _lib.current_stream(), C.byref(eo)))
        o["_keep"] = (iq, is_, ts, ms)
        return o

    def forward_resident(self, iq, is_, ts, ms, edges, off, outputs):
        """Same as forward() on tensors already resident in HBM and pre-packed edges (bench timed region)."""
        o, eo = outputs
        _lib.check(self.lib.ec_forward(self.h, iq.data_ptr(), self._ptr_array(is_), self._ptr_array(ts), ms.data_ptr(),
                                       edges.ctypes.data, off.ctypes.data, iq.shape[0], len(is_), _lib.current_stream(),
                                       C.byref(eo)))
        return o

    def forward_pipelined(self, iq, is_, ts, ms, edges, off, outputs):
        """ec_forward_pipelined on resident tensors: as forward_resident, but the head of this call stays in flight on the library's own
        streams and overlaps the next call's backbone.  `outputs` (a pair from _outputs()) must not be shared with the next call if its
        results are read after that call was enqueued; ALL of them are complete after a pipeline_flush() on the reading stream issued
        before the next pipelined call (include/edgecape_hip.h)."""
        o, eo = outputs
        _lib.check(self.lib.ec_forward_pipelined(self.h, iq.data_ptr(), self._ptr_array(is_), self._ptr_array(ts), ms.data_ptr(),
                                                 edges.ctypes.data, off.ctypes.data, iq.shape[0], len(is_), _lib.current_stream(),
                                                 C.byref(eo)))
        return o

    def pipeline_flush(self, stream=None):
        """Make `stream` (default: torch's current stream) wait for the head of the most recent pipelined call; no host sync."""
        st = _lib.current_stream() if stream is None else stream.cuda_stream
        _lib.check(self.lib.ec_pipeline_flush(self.h, st))

    # ---- support-side episode cache (include/edgecape_hip.h: ec_support_*) ---------------------------
    def support_encode(self, img_s, target_s, mask_s, skeletons, cache=None):
        """Run the support side once per episode: support backbone features, pooled support tokens, SkeletonPredictor.
        img_s / target_s: lists over shots of [n_episodes, ...]; returns an opaque cache for forward_cached()."""
        is_ = [self._dev(x) for x in img_s]
        ts = [self._dev(t) for t in target_s]
        n, S = is_[0].shape[0], len(is_)
        ms = self._dev(mask_s).reshape(n, self.K)
        edges, off = self._edges(skeletons, n)
        if cache is None:
            cache = SupportCache(self, max(self.max_batch, n))
        _lib.check(self.lib.ec_support_encode(self.h, cache.h, self._ptr_array(is_), self._ptr_array(ts), ms.data_ptr(),
                                              edges.ctypes.data, off.ctypes.data, n, S, _lib.current_stream()))
        cache.n_episodes = n
        cache._keep = (is_, ts, ms)
        return cache

    def forward_cached(self, img_q, cache, episode_of_query):
        """Query side only: query b is matched against cached episode episode_of_query[b]."""
        iq = self._dev(img_q)
        bs = iq.shape[0]
        ep = np.ascontiguousarray(np.asarray(episode_of_query, np.int32).reshape(bs))
        o, eo = self._outputs(bs)
        _lib.check(self.lib.ec_forward_cached(self.h, cache.h, iq.data_ptr(), ep.ctypes.data, bs, _lib.current_stream(), C.byref(eo)))
        o["_keep"] = (iq,)
        return o

    def support_cache(self, max_episodes):
        """An empty episode cache of max_episodes slots for support_encode / forward_cached / forward_episodes."""
        return SupportCache(self, max_episodes)

    def prepare_episode_call(self, img_q, slot_of_query, new=None):
        """Arguments of one ec_forward_episodes call, made resident / packed once (the timed loops re-issue prepared calls):
        `new` = dict(img_s, target_s, mask_s, skeletons, slots), lists over shots of [n_new, ...], or None; img_q [bs,3,H,W] or None."""
        keep = []
        if new is not None:
            is_ = [self._dev(x) for x in new["img_s"]]
            ts = [self._dev(t) for t in new["target_s"]]
            n, S = is_[0].shape[0], len(is_)
            ms = self._dev(new["mask_s"]).reshape(n, self.K)
            edges, off = self._edges(new["skeletons"], n)
            slots = np.ascontiguousarray(np.asarray(new["slots"], np.int32).reshape(n))
            pi, pt = self._ptr_array(is_), self._ptr_array(ts)
            keep += [is_, ts, ms, edges, off, slots, pi, pt]
            a_new = (pi, pt, ms.data_ptr(), edges.ctypes.data, off.ctypes.data, slots.ctypes.data, n, S)
            top = int(slots.max()) + 1
        else:
            a_new, top = (None, None, None, None, None, None, 0, 0), 0
        if img_q is not None:
            iq = self._dev(img_q)
            bs = iq.shape[0]
            sq = np.ascontiguousarray(np.asarray(slot_of_query, np.int32).reshape(bs))
            keep += [iq, sq]
            a_q = (iq.data_ptr(), sq.ctypes.data, bs)
        else:
            a_q = (None, None, 0)
        return dict(args=a_new + a_q, bs=a_q[2], top=top, keep=keep)

    def forward_episodes(self, cache, img_q=None, slot_of_query=None, new=None, outputs=None, pipelined=False, prepared=None):
        """ec_forward_episodes (include/edgecape_hip.h): encode the episodes that start in this call into their cache slots and run
        the query side for the call's queries, query b against slot slot_of_query[b]; support and query images share ONE backbone
        pass.  Arguments as prepare_episode_call (or its result as `prepared`).  pipelined: ec_forward_pipelined's completion rule
        (pipeline_flush); `outputs` as for forward_pipelined (default: a fresh set)."""
        p = prepared if prepared is not None else self.prepare_episode_call(img_q, slot_of_query, new)
        if p["bs"] > 0:
            o, eo = outputs if outputs is not None else self._outputs(p["bs"])
        else:
            o, eo = {}, None
        _lib.check(self.lib.ec_forward_episodes(self.h, cache.h, *p["args"], _lib.current_stream(),
                                                C.byref(eo) if eo is not None else None, 1 if pipelined else 0))
        cache.n_episodes = max(cache.n_episodes, p["top"])
        if outputs is None and p["bs"] > 0:
            o["_keep"] = p["keep"]
        cache._keep = p["keep"]
        return o

    def debug(self, name):
        n = C.c_int64()
        _lib.check(self.lib.ec_debug_read(self.h, name.encode(), None, 0, C.byref(n)))
        buf = np.empty(n.value, np.float32)
        _lib.check(self.lib.ec_debug_read(self.h, name.encode(), buf.ctypes.data, n.value, C.byref(n)))
        return buf

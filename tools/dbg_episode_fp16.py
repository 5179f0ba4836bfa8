"""debug: where do ec_forward_episodes and ec_forward differ in fp16 / mixed?  (taps: feature_q, support_keypoints, enc)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine, SupportCache
arch, H, bs, S = "dinov2_vits14", 224, 6, 1
sd = synth.make_weights(arch, seed=61)
sup = synth.make_pairs(2, S, H, seed=300, fixed_n_kp=False)
qry = synth.make_pairs(bs, 1, H, seed=400)
mask = sup["target_weight_s"][0].copy()
skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
ep = np.array([0, 0, 0, 1, 1, 1], np.int32)
for prec, hp in (("fp16", "mixed"), ("fp16", "bf16x3"), ("fp32", "fp32")):
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision=prec, head_precision=hp)
    cache = SupportCache(eng, 3)
    new = dict(img_s=sup["img_s"], target_s=sup["target_s"], mask_s=mask, skeletons=skels, slots=[0, 1])
    o = eng.forward_episodes(cache, qry["img_q"], ep, new=new)
    torch.cuda.synchronize()
    taps_e = {k: eng.debug(k).copy() for k in ("feature_q", "enc")}
    got = {k: v.cpu().numpy() for k, v in o.items() if not k.startswith("_")}
    r = eng.forward(qry["img_q"], [x[ep] for x in sup["img_s"]], [x[ep] for x in sup["target_s"]], mask[ep], [skels[e] for e in ep])
    torch.cuda.synchronize()
    taps_f = {k: eng.debug(k).copy() for k in ("feature_q", "enc")}
    n = min(taps_e["feature_q"].size, taps_f["feature_q"].size)
    print(prec, hp, "feature_q", np.abs(taps_e["feature_q"][:n] - taps_f["feature_q"][:n]).max(), "enc", np.abs(taps_e["enc"] - taps_f["enc"]).max(),
          {k: float(np.abs(got[k] - r[k].cpu().numpy()).max()) for k in got})
    # the backbone alone: the same 6 query images in batches of 6 and of 8 (two more images behind them)
    a = eng.backbone(qry["img_q"], nchw=False).cpu().numpy()
    b = eng.backbone(np.concatenate([qry["img_q"], sup["img_s"][0]], 0), nchw=False).cpu().numpy()[:bs]
    print("   backbone 6 vs 8 images:", np.abs(a - b).max())

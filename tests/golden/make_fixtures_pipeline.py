"""Generates tests/golden/scene_pipeline.npz by RUNNING THE REFERENCE'S OWN DATASET CLASS
(lib/dataset.py: ScannetReferenceDataset.__getitem__) in this container on a tiny synthetic data set written, in the
reference's on-disk formats, to a temporary directory:

    <tmp>/scannet_data/<scene>_aligned_vert.npy   (N x 9 f32: xyz, rgb, normal)
                       <scene>_ins_label.npy / _sem_label.npy   (N,)
                       <scene>_aligned_bbox.npy   (M x 8: centre, size, nyu40 id, object id)
                       <scene>_x.npy / _y.npy / _z.npy   (M x M relation classes)
    <tmp>/glove.p, ScanRefer_filtered.json (unused by the class but part of the layout)

Nothing of the reference is copied and nothing is written under /root/reference (the class writes its vocabulary
files into CONF.PATH.DATA = <tmp>).  Shims: the easydict stand-in of make_fixtures.py and empty ``h5py`` / ``plyfile`` / ``trimesh`` /
``matplotlib`` modules (imported at module level by the reference, never used on this path).  The fixture holds the synthetic inputs, the numpy seeds and the reference's
outputs; ``oracle/scene_pipeline_ref.py`` must reproduce them from (inputs, seed).

Run:  python tests/golden/make_fixtures_pipeline.py
"""
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

NYU_OK = [3, 4, 5, 7, 8, 12, 14, 24, 33, 39]   # nyu40 ids inside DC.nyu40ids
NAMES = {3: "cabinet", 4: "bed", 5: "chair", 7: "table", 8: "door", 12: "counter", 14: "desk", 24: "refrigerator",
         33: "toilet", 39: "otherfurniture"}


def synth_scene(rng, n_vert, n_obj):
    """Vertices on the faces of a room + inside object boxes; instance 0 / semantic 1, 2 = structure."""
    xyz = np.concatenate([rng.uniform(-3, 3, (n_vert, 2)), rng.uniform(0, 2.5, (n_vert, 1))], 1)
    ins = np.zeros(n_vert, dtype=np.int64)
    sem = rng.choice([1, 2, 22], n_vert).astype(np.int64)
    boxes = []
    per = n_vert // (2 * n_obj)
    for o in range(n_obj):
        c = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5), rng.uniform(0.3, 1.5)])
        sz = rng.uniform(0.3, 1.2, 3)
        lo = o * per
        pts = c + (rng.uniform(-0.5, 0.5, (per, 3)) * sz)
        xyz[lo:lo + per] = pts
        ins[lo:lo + per] = o + 1
        nyu = NYU_OK[o % len(NYU_OK)] if o % 5 != 4 else 1   # every fifth object: a structure class (no votes)
        sem[lo:lo + per] = nyu
        boxes.append(list(0.5 * (pts.min(0) + pts.max(0))) + list(pts.max(0) - pts.min(0)) + [NYU_OK[o % len(NYU_OK)], o + 1])
    vert = np.concatenate([xyz, rng.uniform(0, 255, (n_vert, 3)), rng.normal(size=(n_vert, 3))], 1).astype(np.float32)
    rel = [rng.integers(0, 3, (n_obj, n_obj)).astype(np.uint32) for _ in range(3)]
    return vert, ins, sem, np.array(boxes, dtype=np.float64), rel


def main():
    tmp = tempfile.mkdtemp(prefix="spacap_pipeline_")
    os.makedirs(os.path.join(tmp, "scannet_data"))
    rng = np.random.default_rng(7)
    scenes = {"scene0000_00": synth_scene(rng, 5000, 9), "scene0001_00": synth_scene(rng, 3500, 6)}
    for sid, (vert, ins, sem, box, rel) in scenes.items():
        base = os.path.join(tmp, "scannet_data", sid)
        np.save(base + "_aligned_vert.npy", vert)
        np.save(base + "_ins_label.npy", ins)
        np.save(base + "_sem_label.npy", sem)
        np.save(base + "_aligned_bbox.npy", box)
        for a, r in zip("xyz", rel):
            np.save(base + f"_{a}.npy", r)
    words = ["the", "chair", "is", "next", "to", "a", "table", "door", "left", "of", "bed", "white", "brown", "near"]
    glove = {w: rng.normal(size=300) for w in words + ["unk", "sos", "eos"]}
    pickle.dump(glove, open(os.path.join(tmp, "glove.p"), "wb"))
    scanrefer = []
    for sid, (_, _, _, box, _) in scenes.items():
        for k in range(3):
            o = int(box[(2 * k + 1) % len(box), 7])
            n = 4 + 3 * k
            toks = [words[(o + i * (k + 1)) % len(words)] for i in range(n)] + (["zebra"] if k == 1 else [])
            scanrefer.append({"scene_id": sid, "object_id": str(o), "object_name": NAMES[int(box[o - 1, 6])],
                              "ann_id": str(k), "token": toks})
    json.dump(scanrefer, open(os.path.join(tmp, "ScanRefer_filtered.json"), "w"))

    # ---- import the reference with its paths pointed at <tmp> (meta data stays the reference's own, read-only) ----
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    # h5py is not installed here.  The reference opens its multiview database with h5py.File(path, "r", libver="latest")
    # and indexes it by scene id (lib/dataset.py:321-327): the stand-in serves the same rows from an .npz written below.
    h5 = types.ModuleType("h5py")

    class _File:
        def __init__(self, path, *a, **k):
            self._z = np.load(path)

        def __getitem__(self, key):
            return self._z[key]

    h5.File = _File
    sys.modules["h5py"] = h5
    # utils/pc_utils.py imports PLY / mesh / plotting packages at module level (unused on this path)
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = object
    sys.modules["plyfile"] = ply
    for name in ("trimesh", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    from lib.config import CONF
    CONF.PATH.SCANNET = os.path.join(REF, "data", "scannet")
    CONF.PATH.SCANNET_META = os.path.join(REF, "data", "scannet", "meta_data")
    CONF.PATH.DATA = tmp
    CONF.PATH.SCANNET_DATA = os.path.join(tmp, "scannet_data")
    # multiview rows (128 per vertex): integer-hash values, regenerated identically by the tests (not stored)
    sys.path.insert(0, HERE)
    from detweights import _uniform
    mv = {sid: (_uniform(v[0].shape[0] * 128, "multiview_" + sid, 5) + np.float32(0.5)).reshape(-1, 128) for sid, v in scenes.items()}
    np.savez(os.path.join(tmp, "enet_feats_maxpool.npz"), **mv)
    CONF.MULTIVIEW = os.path.join(tmp, "enet_feats_maxpool.npz")
    from lib.dataset import ScannetReferenceDataset

    num_points = 4096
    ds = ScannetReferenceDataset(scanrefer=scanrefer, split="train", name="ScanRefer", num_points=num_points,
                                 use_height=True, use_color=False, use_normal=False, use_multiview=False,
                                 augment=True, use_relation=True)
    out = {"num_points": num_points, "n_items": len(scanrefer)}
    for sid, (vert, ins, sem, box, rel) in scenes.items():
        out[f"{sid}/vert"], out[f"{sid}/ins"], out[f"{sid}/sem"], out[f"{sid}/bbox"] = vert, ins, sem, box
        for a, r in zip("xyz", rel):
            out[f"{sid}/{a}"] = r
    out["scene_ids"] = np.array(list(scenes))
    out["item_scene"] = np.array([d["scene_id"] for d in scanrefer])
    out["item_object"] = np.array([int(d["object_id"]) for d in scanrefer])
    out["item_object_name"] = np.array([d["object_name"] for d in scanrefer])
    out["item_ann"] = np.array([int(d["ann_id"]) for d in scanrefer])
    out["item_tokens"] = np.array(["|".join(d["token"]) for d in scanrefer])
    out["glove_words"] = np.array(list(glove))
    out["glove_vecs"] = np.stack([glove[w] for w in glove])
    from lib.dataset import DC
    out["mean_size_arr"] = DC.mean_size_arr
    out["nyu40id2class_keys"] = np.array(list(DC.nyu40id2class.keys()))
    out["nyu40id2class_vals"] = np.array(list(DC.nyu40id2class.values()))
    out["raw2label_names"] = np.array(sorted(set(d["object_name"] for d in scanrefer)))
    out["raw2label_vals"] = np.array([ds.raw2label.get(n, 17) for n in out["raw2label_names"]])
    out["vocab_words"] = np.array(list(ds.vocabulary["word2idx"]))
    out["vocab_ids"] = np.array([ds.vocabulary["word2idx"][w] for w in ds.vocabulary["word2idx"]])
    # items are drawn in order; the class mutates the scene's x / y relation labels on every flip (lib/dataset.py:
    # 369-384), so the order is part of the fixture
    order = [0, 3, 1, 4, 2, 5, 0]
    out["order"] = np.array(order)
    for step, idx in enumerate(order):
        seed = 1000 + 17 * step
        np.random.seed(seed)
        d = ds[idx]
        out[f"step{step}/seed"] = seed
        for k, v in d.items():
            if k == "load_time":
                continue
            out[f"step{step}/{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "scene_pipeline.npz"), **out)
    print("wrote scene_pipeline.npz:", len(out), "arrays;", {k: np.asarray(v).shape for k, v in d.items() if k != "load_time"})

    # ---- second fixture: the extra input channels (BASELINE configs 3 and 4) -------------------------------------------
    # colour (with the reference's re-normalisation of the cached scene on every access, lib/dataset.py:312-315), normals
    # and the 128 multiview channels.  Inputs are those of the first fixture + the hashed multiview rows; per step only
    # the point cloud rows ::8 (all channels) and the colours are kept.
    feats = {}
    for tag, kw in (("color_normal", dict(use_color=True, use_normal=True, use_multiview=False)),
                    ("multiview_normal", dict(use_color=False, use_normal=True, use_multiview=True)),
                    ("all", dict(use_color=True, use_normal=True, use_multiview=True))):
        ds2 = ScannetReferenceDataset(scanrefer=scanrefer, split="train", name="ScanRefer", num_points=1024, use_height=True,
                                      augment=True, use_relation=True, **kw)
        for step, idx in enumerate([0, 3, 0, 1, 0]):      # scene 0 is visited four times: four different colour states
            seed = 2000 + 13 * step
            np.random.seed(seed)
            d = ds2[idx]
            feats[f"{tag}/step{step}/seed"] = seed
            feats[f"{tag}/step{step}/idx"] = idx
            feats[f"{tag}/step{step}/point_clouds_rows8"] = np.asarray(d["point_clouds"])[::8]
            feats[f"{tag}/step{step}/pcl_color"] = np.asarray(d["pcl_color"])
            feats[f"{tag}/step{step}/vote_label_mask"] = np.asarray(d["vote_label_mask"])
    np.savez_compressed(os.path.join(HERE, "scene_pipeline_feats.npz"), **feats)
    print("wrote scene_pipeline_feats.npz:", len(feats), "arrays; channels", {t: feats[f"{t}/step0/point_clouds_rows8"].shape for t in ("color_normal", "multiview_normal", "all")})


if __name__ == "__main__":
    main()

"""Input pipeline on the device: scenes in the reference's on-disk formats are loaded
once into HBM; a training batch -- random 40 000-point subsample, height feature, flips / rotations / translation,
vote labels, box / class / relation labels -- is then produced by two HIP kernels + batched torch ops on <= 128
boxes, with no per-item host work on the points.

Mirror of ``ScannetReferenceDataset`` (lib/dataset.py:247-531) for the ScanRefer training path: ``use_height`` /
``use_normal`` / ``use_color`` / ``use_multiview`` / ``augment`` / ``use_relation``, output channels in the reference's
order xyz, [rgb], [normal], [multiview 128], [height] (:309-333).  Colour: the reference's
``point_cloud[:, 3:6] = (point_cloud[:, 3:6] - MEAN_COLOR_RGB) / 256.0`` writes through a VIEW into the cached scene
(:312-315), so a scene's colours are normalised once more on every access -- within one DataLoader worker and one epoch,
because its workers are re-forked from an unmodified parent every epoch.  ``color_renorm="once"`` (the default) normalises a
scene's colours once, when it is added; ``color_renorm="per_access"`` reproduces the quirk (the per-scene colour tensor in HBM
is re-normalised in place per visit, item by item: float64 arithmetic, float32 store, as numpy does) for parity with the
reference's items, and ``reset_colors()`` restores the loaded colours (call it at epoch start to get the reference's per-epoch
reset instead of colours that collapse towards -mean / 256 for the rest of training).  Multiview rows
(``enet_feats_maxpool.hdf5``: N x 128 float32 per scene, :321-328) are handed over per scene and stay resident: all
1 513 scenes are ~116 GB of the 288 GB.
Same keys, dtypes and shapes as the reference's ``data_dict`` with a leading batch dimension (what its DataLoader's
default collate produces).  A 288 GB MI355X holds all 1 513 ScanNet scenes (~150 k vertices x 40 B = 9 GB) many times
over, so nothing is re-read from disk after start-up.

Reference quirk kept on purpose: an x / y flip swaps the classes 0 <-> 2 of the scene's cached relation matrix in
place (:369-384), so the relation labels of a scene depend on the history of flips it has drawn.  ``flip_parity``
holds that state here.

Randomness: ``draw()`` produces the per-item random numbers (subsample, flips, angles, translation) from a torch
generator; ``batch(..., draws=...)`` accepts externally supplied draws, which is how the parity tests feed the
reference's own numpy draws (oracle/scene_pipeline_ref.draws_from_seed).
"""
import math
import os

import numpy as np
import torch

from ._native import check, lib

MAX_NUM_OBJ = 128          # lib/dataset.py:26
MAX_DES_LEN = 30           # lib/config.py: CONF.TRAIN.MAX_DES_LEN
NYU40IDS = (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
            34, 35, 36, 37, 38, 39, 40)   # data/scannet/model_util_scannet.py:88


def _rot(axis, t):
    """rotx / roty / rotz of utils/pc_utils.py:282-320 (float64)."""
    c, s = math.cos(t), math.sin(t)
    return {"x": [[1, 0, 0], [0, c, -s], [0, s, c]], "y": [[c, 0, s], [0, 1, 0], [-s, 0, c]],
            "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]


class DeviceSceneDataset:
    def __init__(self, device, mean_size_arr, nyu40id2class, raw2label=None, num_points=40000, use_height=True,
                 use_normal=False, augment=True, use_relation=True, max_instances=1024, use_color=False,
                 use_multiview=False, color_renorm="once"):
        if color_renorm not in ("once", "per_access"):
            raise ValueError("color_renorm must be 'once' or 'per_access'")
        self._color_renorm = color_renorm   # fixed at construction: which copies a scene keeps in HBM depends on it (add_scene)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CPU not supported")
        self.num_points, self.use_height, self.use_normal = num_points, use_height, use_normal
        self.augment, self.use_relation, self.max_instances = augment, use_relation, max_instances
        self.mean_size = torch.as_tensor(np.asarray(mean_size_arr), dtype=torch.float64, device=self.device)
        lut = torch.zeros(64, dtype=torch.int64)
        for k, v in dict(nyu40id2class).items():
            lut[int(k)] = int(v)
        self.nyu2class = lut.to(self.device)
        self.raw2label = dict(raw2label or {})
        self.use_color, self.use_multiview = use_color, use_multiview
        self.Cn = 3 + (3 if use_normal else 0) + (1 if use_height else 0)      # channels of the per-scene "feat" rows
        self.C = self.Cn + (3 if use_color else 0) + (128 if use_multiview else 0)
        # destination channel of every "feat" channel in the output rows: xyz, [rgb], [normal], [multiview], [height]
        off_n = 3 + (3 if use_color else 0)
        off_mv = off_n + (3 if use_normal else 0)
        off_h = off_mv + (128 if use_multiview else 0)
        self._off_color, self._off_mv = 3, off_mv
        m = [0, 1, 2] + ([off_n, off_n + 1, off_n + 2] if use_normal else []) + ([off_h] if use_height else [])
        self._dst_off = torch.tensor(m, dtype=torch.int32, device=self.device) if self.C != self.Cn else None
        self._mean_rgb = torch.tensor([109.8, 97.2, 83.8], dtype=torch.float64, device=self.device)   # lib/dataset.py:28
        self._scene_index = {}
        self._scenes = []       # per scene: the big per-vertex tensors (kept alive here; the kernels get pointers)
        self._items = []
        self._tables = None     # stacked per-scene / per-item label tables, built lazily by _finalize()

    def _loaded_color(self, vert):
        col = torch.as_tensor(np.ascontiguousarray(vert[:, 3:6], dtype=np.float32)).to(self.device)
        if self.use_color and self.color_renorm == "once":   # (without use_color the reference never normalises: pcl_color stays raw)
            col = ((col.double() - self._mean_rgb) / 256.0).float()
        return col

    @property
    def color_renorm(self):
        """Read-only: "once" scenes are normalised when added and keep no raw copy, "per_access" scenes keep ``color0`` for
        reset_colors(); switching the mode of a loaded dataset would leave scenes without the copy the other mode needs."""
        return self._color_renorm

    def reset_colors(self):
        """``color_renorm="per_access"``: back to the colours as loaded (the reference's state at the start of every epoch)."""
        if self.color_renorm == "per_access":
            for sc in self._scenes:
                sc["color"].copy_(sc["color0"])

    # ---- loading -----------------------------------------------------------------------------------------------
    def add_scene(self, scene_id, vert, ins, sem, bbox, x=None, y=None, z=None, multiview=None):
        """Arrays as the reference loads them (lib/dataset.py:197-212): vert (N,9) f32, ins / sem (N,), bbox (M,8)
        [centre, size, nyu40 id, object id], x / y / z (M,M) relation classes; ``multiview``: the scene's (N,128)
        float32 rows of the multiview database (required with use_multiview)."""
        vert = np.asarray(vert)
        if self.use_multiview:
            if multiview is None or tuple(np.shape(multiview)) != (vert.shape[0], 128):
                raise ValueError("use_multiview needs the scene's (N, 128) multiview rows")
        pc = vert[:, 0:3]
        if self.use_normal:
            pc = np.concatenate([pc, vert[:, 6:9]], 1)
        if self.use_height:   # one-off per scene: the 0.99th percentile of z is the floor (:330-333)
            floor = np.percentile(pc[:, 2], 0.99)
            pc = np.concatenate([pc, np.expand_dims(pc[:, 2] - floor, 1)], 1)
        if int(np.asarray(ins).max(initial=0)) >= self.max_instances:
            raise ValueError("instance label >= max_instances")
        d = self.device
        bbox = np.asarray(bbox, dtype=np.float64)
        nb = min(bbox.shape[0], MAX_NUM_OBJ)
        box8 = np.zeros((MAX_NUM_OBJ, 8))
        box8[:nb] = bbox[:nb]
        rel = np.zeros((3, MAX_NUM_OBJ, MAX_NUM_OBJ), dtype=np.int64)
        for k, r in enumerate((x, y, z)):
            if r is not None:
                rel[k, :nb, :nb] = np.asarray(r)[:nb, :nb]
        self._scene_index[scene_id] = len(self._scenes)
        self._scenes.append({
            "feat": torch.as_tensor(np.ascontiguousarray(pc, dtype=np.float32)).to(d),
            "color": self._loaded_color(vert),
            "ins": torch.as_tensor(np.asarray(ins).astype(np.int32)).to(d),
            "isobj": torch.as_tensor(np.isin(np.asarray(sem), NYU40IDS).astype(np.uint8)).to(d),
            "n": int(vert.shape[0]), "nb": nb, "box8": box8, "rel": rel,
        })
        if self.color_renorm == "per_access":   # the raw copy only exists where reset_colors() needs it (~1 GB over all of ScanNet)
            self._scenes[-1]["color0"] = torch.as_tensor(np.ascontiguousarray(vert[:, 3:6], dtype=np.float32)).to(d)
        if self.use_multiview:
            self._scenes[-1]["multiview"] = torch.as_tensor(np.ascontiguousarray(multiview, dtype=np.float32)).to(d)
        self._tables = None

    def load_scene(self, root, scene_id, multiview_db=None):
        """``<root>/<scene_id>_aligned_vert.npy`` etc., the layout of CONF.PATH.SCANNET_DATA.  ``multiview_db``: an
        open mapping scene_id -> (N,128) rows (the reference's h5py.File on enet_feats_maxpool.hdf5, CONF.MULTIVIEW)."""
        base = os.path.join(root, scene_id)
        rel = [np.load(base + f"_{a}.npy") if os.path.exists(base + f"_{a}.npy") else None for a in "xyz"]
        mv = np.asarray(multiview_db[scene_id]) if (self.use_multiview and multiview_db is not None) else None
        self.add_scene(scene_id, np.load(base + "_aligned_vert.npy"), np.load(base + "_ins_label.npy"),
                       np.load(base + "_sem_label.npy"), np.load(base + "_aligned_bbox.npy"), *rel, multiview=mv)

    def add_item(self, scene_id, object_id, object_name="", lang_feat=None, lang_ids=None, lang_len=None, ann_id=0):
        """One ScanRefer description (an entry of ``self.scanrefer``).  ``lang_feat`` (32,300) / ``lang_ids`` (32,)
        are the arrays ``_tranform_des`` builds (lib/dataset.py:76-118)."""
        it = {"scene": self._scene_index[scene_id], "object_id": int(object_id),
              "object_cat": int(self.raw2label.get(object_name, 17)), "ann_id": int(ann_id)}
        if lang_ids is not None:
            it["lang_ids"] = np.asarray(lang_ids).astype(np.int64)
            it["lang_feat"] = np.asarray(lang_feat, dtype=np.float32)
            it["lang_len"] = int(lang_len)
        self._items.append(it)
        self._tables = None
        return len(self._items) - 1

    def __len__(self):
        return len(self._items)

    def _finalize(self):
        if self._tables is not None:
            return self._tables
        d = self.device
        sc, it = self._scenes, self._items
        t = {
            "box8": torch.as_tensor(np.stack([s["box8"] for s in sc])).to(d),                       # (S,128,8) f64
            "nb": torch.tensor([s["nb"] for s in sc], dtype=torch.int64, device=d),
            "rel": torch.as_tensor(np.stack([s["rel"] for s in sc])).to(d),                         # (S,3,128,128) i64
            "nvert": torch.tensor([s["n"] for s in sc], dtype=torch.int64, device=d),
            "ptrs": torch.tensor([[s[k].data_ptr() for k in ("feat", "ins", "isobj", "color")] +
                                  [s["multiview"].data_ptr() if "multiview" in s else 0] for s in sc],
                                 dtype=torch.int64, device=d),                                     # (S,5)
            "parity": torch.zeros(len(sc), 2, dtype=torch.int64, device=d),                         # flip state (x, y)
            "item_scene": torch.tensor([i["scene"] for i in it], dtype=torch.int64, device=d),
            "item_object": torch.tensor([i["object_id"] for i in it], dtype=torch.float64, device=d),
            "item_cat": torch.tensor([i["object_cat"] for i in it], dtype=torch.int64, device=d),
            "item_ann": torch.tensor([i["ann_id"] for i in it], dtype=torch.int64, device=d),
            "nmax": max(s["n"] for s in sc),
        }
        if it and all("lang_ids" in i for i in it):
            ids = torch.as_tensor(np.stack([i["lang_ids"] for i in it]))
            t["lang_ids"] = ids.to(d)
            t["lang_label"] = torch.cat([torch.ones(len(it), 1, dtype=torch.int64), ids], 1).to(d)   # :479-481
            t["lang_feat"] = torch.as_tensor(np.stack([i["lang_feat"] for i in it])).to(d)
            t["lang_len"] = torch.tensor([i["lang_len"] for i in it], dtype=torch.int64, device=d)
        old = getattr(self, "_parity_keep", None)
        if old is not None and old.shape == t["parity"].shape:
            t["parity"] = old
        self._tables = t
        return t

    # ---- randomness --------------------------------------------------------------------------------------------
    def draw(self, indices, generator=None):
        """Per-item random numbers in the reference's roles (lib/dataset.py:335, 366-401, 233-235), drawn ON THE
        DEVICE without a host sync: (choices (B,P) int32, aug (B,32) float64) for ``batch(..., draws=...)``.
        ``generator``: a torch.Generator of this device (None = the default one)."""
        t = self._finalize()
        dev, P = self.device, self.num_points
        idx = torch.as_tensor(list(indices), dtype=torch.int64, device=dev)
        B = idx.numel()
        n = t["nvert"][t["item_scene"][idx]]                                              # (B,)
        u = torch.rand(B, t["nmax"], device=dev, generator=generator)
        u = torch.where(torch.arange(t["nmax"], device=dev)[None, :] < n[:, None], u, torch.full_like(u, 2.0))
        order = u.argsort(1)
        if t["nmax"] >= P:
            without = order[:, :P]
        else:
            without = torch.zeros(B, P, dtype=torch.int64, device=dev)
        with_rep = (torch.rand(B, P, device=dev, generator=generator, dtype=torch.float64) * n[:, None]).long()
        with_rep = torch.minimum(with_rep, n[:, None] - 1)
        choices = torch.where((n < P)[:, None], with_rep, without).to(torch.int32)       # replace iff N < P (:35)
        aug = torch.zeros(B, int(lib.spacap_scene_aug_doubles()), dtype=torch.float64, device=dev)
        if self.augment:
            r = torch.rand(B, 5, device=dev, generator=generator, dtype=torch.float64)
            aug[:, 0:2] = (r[:, 0:2] > 0.5).double()
            ang = r[:, 2:5] * math.pi / 18 - math.pi / 36
            c, s = torch.cos(ang), torch.sin(ang)
            one, zero = torch.ones(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev)
            Rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1)
            Ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1)
            Rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1)
            aug[:, 2:11], aug[:, 11:20], aug[:, 20:29] = Rx, Ry, Rz
            k = torch.randint(0, 1001, (B, 3), device=dev, generator=generator)
            aug[:, 29:32] = -0.5 + k.double() * 0.001                                       # np.arange(-0.5, 0.501, 0.001)[k]
        return choices, aug

    def _host_draws(self, draws):
        """Draws given as the reference's numbers (dicts: choices, flip_x, flip_y, angles, translation) -> tensors."""
        dev = self.device
        nd = int(lib.spacap_scene_aug_doubles())
        aug_h = np.zeros((len(draws), nd))
        if self.augment:
            for b, d in enumerate(draws):
                aug_h[b, 0], aug_h[b, 1] = float(d["flip_x"]), float(d["flip_y"])
                for r, (ax, t) in enumerate(zip("xyz", d["angles"])):
                    aug_h[b, 2 + 9 * r: 11 + 9 * r] = np.array(_rot(ax, t), dtype=np.float64).reshape(-1)
                aug_h[b, 29:32] = d["translation"]
        choices = torch.as_tensor(np.stack([np.asarray(d["choices"]) for d in draws]).astype(np.int32)).to(dev)
        return choices, torch.as_tensor(aug_h).to(dev)

    # ---- batch -------------------------------------------------------------------------------------------------
    def batch(self, indices, draws=None, generator=None):
        """data_dict of device tensors for the descriptions ``indices`` (in order, as the reference's single-process
        loader would process them: the relation-label flip state advances item by item).  ``draws``: None (drawn on
        the device), the (choices, aug) pair of ``draw()``, or a list of per-item dicts with the reference's numbers.
        No host synchronisation when the draws come from the device."""
        t = self._finalize()
        dev, P, C = self.device, self.num_points, self.C
        if draws is None:
            choices, aug = self.draw(indices, generator)
        elif isinstance(draws, (tuple,)):
            choices, aug = draws
        else:
            choices, aug = self._host_draws(draws)
        idx = torch.as_tensor(list(indices), dtype=torch.int64, device=dev)
        B = idx.numel()
        sidx = t["item_scene"][idx]
        st = torch.cuda.current_stream(dev).cuda_stream
        f64 = dict(dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            ptrs = t["ptrs"][sidx].t().contiguous()                                        # (5,B) device pointers
            color_ptrs = ptrs[3]
            if self.use_color and self.color_renorm == "per_access":
                # the reference normalises the CACHED colours of the scene on every access (see the module docstring):
                # visit the items in order; an item whose scene comes up again later in this batch keeps a snapshot
                order = [int(i) for i in indices]
                scene_of = [self._items[i]["scene"] for i in order]
                keep, cp = [], []
                for b, sc_i in enumerate(scene_of):
                    col = self._scenes[sc_i]["color"]
                    col.copy_(((col.double() - self._mean_rgb) / 256.0).float())
                    if sc_i in scene_of[b + 1:]:
                        col = col.clone()
                        keep.append(col)
                    cp.append(col.data_ptr())
                color_ptrs = torch.tensor(cp, dtype=torch.int64, device=dev)
            pc = torch.empty(B, P, C, dtype=torch.float32, device=dev)
            ins = torch.empty(B, P, dtype=torch.int32, device=dev)
            isobj = torch.empty(B, P, dtype=torch.uint8, device=dev)
            color = torch.empty(B, P, 3, dtype=torch.float32, device=dev)
            check(lib.spacap_scene_sample_augment_map_f32(
                ptrs[0].data_ptr(), ptrs[1].data_ptr(), ptrs[2].data_ptr(), color_ptrs.data_ptr(), choices.data_ptr(),
                aug.data_ptr(), B, P, self.Cn, int(self.augment), C, self._dst_off.data_ptr() if self._dst_off is not None else None,
                pc.data_ptr(), ins.data_ptr(), isobj.data_ptr(), color.data_ptr(), st), "spacap_scene_sample_augment_map_f32")
            if self.use_color:      # rgb columns = the (normalised) colours the kernel just gathered into pcl_color
                check(lib.spacap_scene_gather_rows_f32(color_ptrs.data_ptr(), choices.data_ptr(), B, P, 3, pc.data_ptr(), C,
                                                       self._off_color, st), "spacap_scene_gather_rows_f32")
            if self.use_multiview:
                check(lib.spacap_scene_gather_rows_f32(ptrs[4].data_ptr(), choices.data_ptr(), B, P, 128, pc.data_ptr(), C,
                                                       self._off_mv, st), "spacap_scene_gather_rows_f32")
            votes = torch.empty(B, P, 9, dtype=torch.float32, device=dev)
            vmask = torch.empty(B, P, dtype=torch.int64, device=dev)
            ws = torch.empty(int(lib.spacap_scene_votes_workspace_bytes(B, self.max_instances)), dtype=torch.uint8, device=dev)
            check(lib.spacap_scene_votes_f32(pc.data_ptr(), ins.data_ptr(), isobj.data_ptr(), B, P, C, self.max_instances,
                                             ws.data_ptr(), votes.data_ptr(), vmask.data_ptr(), st), "spacap_scene_votes_f32")
            # ---- boxes and labels: <= 128 rows per item, float64 like the reference, batched over the items ----
            box8 = t["box8"][sidx]                                                         # (B,128,8)
            nb = t["nb"][sidx]
            valid = torch.arange(MAX_NUM_OBJ, device=dev)[None, :] < nb[:, None]
            mask = valid.double()
            tb = box8[:, :, 0:6].clone()
            if self.augment:
                fx, fy = aug[:, 0:1] != 0, aug[:, 1:2] != 0
                tb[:, :, 0] = torch.where(fx, -1 * tb[:, :, 0], tb[:, :, 0])
                tb[:, :, 1] = torch.where(fy, -1 * tb[:, :, 1], tb[:, :, 1])
                for r, ax in enumerate("xyz"):
                    tb = _rotate_aligned_boxes(tb, aug[:, 2 + 9 * r: 11 + 9 * r].view(B, 3, 3), ax)
                tb[:, :, 0:3] += aug[:, None, 29:32]
            rel = {}
            if self.use_relation:
                # The reference swaps classes 0 <-> 2 of the scene's CACHED matrix on every x / y flip (:369-384):
                # item b sees the scene's parity after its own flip and after the flips of earlier items of the batch.
                flips = (aug[:, 0:2] != 0).double() if self.augment else torch.zeros(B, 2, **f64)
                same = (sidx[:, None] == sidx[None, :]) & (torch.arange(B, device=dev)[:, None] >= torch.arange(B, device=dev)[None, :])
                par = (t["parity"][sidx].double() + same.double() @ flips).long() % 2              # (B,2)
                t["parity"] = (t["parity"] + torch.zeros_like(t["parity"]).index_add_(0, sidx, flips.long())) % 2
                self._parity_keep = t["parity"]
                region = valid[:, :, None] & valid[:, None, :]
                R3 = t["rel"][sidx]                                                        # (B,3,128,128)
                for k, a in enumerate("xyz"):
                    m = R3[:, k]
                    if k < 2:
                        sw = torch.where(m == 0, 2, torch.where(m == 2, 0, m))
                        m = torch.where((par[:, k] == 1)[:, None, None] & region, sw, m)
                    rel[f"{a}_label"] = m
            cls = torch.where(valid, self.nyu2class[box8[:, :, 6].long().clamp(0, 63)], torch.zeros_like(nb)[:, None])
            size_res = torch.where(valid.unsqueeze(-1), tb[:, :, 3:6] - self.mean_size[cls], torch.zeros_like(tb[:, :, 3:6]))
            ids = torch.where(valid, box8[:, :, 7], torch.zeros_like(mask))
            obj = t["item_object"][idx]
            ref_box = (valid & (ids == obj[:, None])).long()
            # the reference's loop keeps the LAST matching box (:437-449)
            last = (ref_box * torch.arange(1, MAX_NUM_OBJ + 1, device=dev)).argmax(1)
            has = ref_box.sum(1) > 0
            bi = torch.arange(B, device=dev)
            ref_center = torch.where(has[:, None], tb[bi, last, 0:3], torch.zeros(B, 3, **f64))
            ref_cls = torch.where(has, cls[bi, last], torch.zeros(B, dtype=torch.int64, device=dev))
            ref_res = torch.where(has[:, None], size_res[bi, last], torch.zeros(B, 3, **f64))
            size = self.mean_size[cls] + size_res
            corners = torch.where(valid[:, :, None, None], _corners(tb[:, :, 0:3], size), torch.zeros(B, MAX_NUM_OBJ, 8, 3, **f64))
            ref_corners = torch.where(has[:, None, None], corners[bi, last], torch.zeros(B, 8, 3, **f64))
        d = {
            "point_clouds": pc, "pcl_color": color, "center_label": tb[:, :, 0:3].float(),
            "heading_class_label": torch.zeros(B, MAX_NUM_OBJ, dtype=torch.int64, device=dev),
            "heading_residual_label": torch.zeros(B, MAX_NUM_OBJ, dtype=torch.float32, device=dev),
            "size_class_label": cls, "size_residual_label": size_res.float(),
            "num_bbox": nb, "sem_cls_label": cls.clone(),
            "scene_object_ids": ids.long(), "box_label_mask": mask.float(), "box_label_mask_int": mask.long(),
            "vote_label": votes, "vote_label_mask": vmask, "ref_box_label": ref_box, "ref_center_label": ref_center.float(),
            "ref_heading_class_label": torch.zeros(B, dtype=torch.int64, device=dev),
            "ref_heading_residual_label": torch.zeros(B, dtype=torch.int64, device=dev),
            "ref_size_class_label": ref_cls, "ref_size_residual_label": ref_res.float(),
            "ref_box_corner_label": ref_corners, "gt_box_corner_label": corners, "gt_box_masks": mask.long(),
            "gt_box_object_ids": ids.long(), "object_id": obj.long(), "object_cat": t["item_cat"][idx],
            "ann_id": t["item_ann"][idx], "dataset_idx": idx,
        }
        d.update(rel)
        if "lang_ids" in t:
            for k in ("lang_feat", "lang_ids", "lang_label", "lang_len"):
                d[k] = t[k][idx]
        return d


def _rotate_aligned_boxes(tb, R, axis):
    """rotate_aligned_boxes_along_axis (data/scannet/model_util_scannet.py:47-79), batched: tb (B,M,6), R (B,3,3)."""
    centers, lengths = tb[:, :, 0:3], tb[:, :, 3:6]
    Rt = R.transpose(1, 2)
    new_centers = centers @ Rt
    a, b = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[axis]
    d1, d2 = lengths[:, :, a] / 2.0, lengths[:, :, b] / 2.0
    n1, n2 = [], []
    for s1, s2 in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        crn = torch.zeros_like(centers)
        crn[:, :, 0] = s1 * d1
        crn[:, :, 1] = s2 * d2
        crn = crn @ Rt
        n1.append(crn[:, :, 0])
        n2.append(crn[:, :, 1])
    nd1 = 2.0 * torch.stack(n1, -1).max(-1).values
    nd2 = 2.0 * torch.stack(n2, -1).max(-1).values
    out = [lengths[:, :, 0], lengths[:, :, 1], lengths[:, :, 2]]
    out[a], out[b] = nd1, nd2
    return torch.cat([new_centers, torch.stack(out, -1)], -1)


_SIGNS = ((1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1))


def _corners(center, size):
    """get_3d_box_batch (utils/box_util.py:360-383) at heading 0: corner k = centre + signs_k * size / 2."""
    sg = torch.tensor(_SIGNS, dtype=center.dtype, device=center.device)
    return center.unsqueeze(-2) + sg * (size.unsqueeze(-2) / 2)

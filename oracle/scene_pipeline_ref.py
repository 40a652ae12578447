"""CPU restatement (numpy) of the reference's per-item input pipeline -- TEST INFRASTRUCTURE, never imported by
the product (spacap3d_amd/).

Follows ``ScannetReferenceDataset.__getitem__`` (lib/dataset.py:291-531) and its helpers ``random_sampling``
(utils/pc_utils.py:32-40), ``rotx / roty / rotz`` (utils/pc_utils.py:282-296 and rotz), ``rotate_aligned_boxes_along_axis``
(data/scannet/model_util_scannet.py:47-79), ``_translate`` (lib/dataset.py:229-245), ``param2obb_batch``
(model_util_scannet.py:165-172) and ``get_3d_box_batch`` (utils/box_util.py:360-383), including the reference's
quirks that are observable in its outputs:

  * the x / y relation matrices of a scene are flipped IN PLACE (0 <-> 2) whenever an item of that scene draws the
    corresponding flip (lib/dataset.py:369-384), i.e. the labels a scene returns depend on the history of flips;
  * the votes are computed from the SAMPLED, AUGMENTED points (instance boxes = min / max of the sampled points of an
    instance, lib/dataset.py:415-428), an instance votes iff the semantic label of its first sampled point is one of the
    37 object classes;
  * heading is identically 0 (angle classes / residuals are never filled in).

Pinned against the reference itself: ``tests/golden/scene_pipeline.npz`` holds the outputs of the reference's class
run in the build container (tests/golden/make_fixtures_pipeline.py) on synthetic scenes; ``tests/test_scene_pipeline.py``
requires this restatement to reproduce every array bit for bit from (inputs, numpy seed).

The randomness is split off: ``draws_from_seed`` replays numpy's global generator in the reference's call order
and returns the draws as plain numbers, so the device pipeline can be fed the same draws.
"""
import numpy as np

MAX_NUM_OBJ = 128                       # lib/dataset.py:26
MAX_DES_LEN = 30                        # lib/config.py: CONF.TRAIN.MAX_DES_LEN
MEAN_COLOR_RGB = np.array([109.8, 97.2, 83.8])   # lib/dataset.py:28
NYU40IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 28, 29,
                     30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40])   # model_util_scannet.py:88
TYPE2CLASS = {'cabinet': 0, 'bed': 1, 'chair': 2, 'sofa': 3, 'table': 4, 'door': 5, 'window': 6, 'bookshelf': 7,
              'picture': 8, 'counter': 9, 'desk': 10, 'curtain': 11, 'refrigerator': 12, 'shower curtain': 13,
              'toilet': 14, 'sink': 15, 'bathtub': 16, 'others': 17}   # model_util_scannet.py:83-85


def draws_from_seed(seed, n_vert, num_points, augment=True):
    """The random numbers one ``__getitem__`` call consumes, in the reference's order: np.random.choice for the
    subsample (:335, replace iff n_vert < num_points), then (augment only) flip-x, flip-y, three angles (:366-401),
    three translation picks from arange(-0.5, 0.501, 0.001) (:233-235)."""
    rs = np.random.RandomState(seed)
    d = {"choices": rs.choice(n_vert, num_points, replace=(n_vert < num_points))}
    if augment:
        d["flip_x"] = bool(rs.random_sample() > 0.5)
        d["flip_y"] = bool(rs.random_sample() > 0.5)
        d["angles"] = [(rs.random_sample() * np.pi / 18) - np.pi / 36 for _ in range(3)]
        grid = np.arange(-0.5, 0.501, 0.001)
        d["translation"] = [rs.choice(grid, size=1)[0] for _ in range(3)]
    return d


def rotx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rotate_aligned_boxes_along_axis(boxes, rot_mat, axis):
    """model_util_scannet.py:47-79: rotate the centres, new size = bounding extent of the rotated face."""
    centers, lengths = boxes[:, 0:3], boxes[:, 3:6]
    new_centers = np.dot(centers, np.transpose(rot_mat))
    a, b = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[axis]
    d1, d2 = lengths[:, a] / 2.0, lengths[:, b] / 2.0
    new_1 = np.zeros((d1.shape[0], 4))
    new_2 = np.zeros((d1.shape[0], 4))
    for i, (s1, s2) in enumerate([(-1, -1), (1, -1), (1, 1), (-1, 1)]):
        crn = np.zeros((d1.shape[0], 3))
        crn[:, 0] = s1 * d1
        crn[:, 1] = s2 * d2
        crn = np.dot(crn, np.transpose(rot_mat))
        new_1[:, i] = crn[:, 0]
        new_2[:, i] = crn[:, 1]
    nd1, nd2 = 2.0 * np.max(new_1, 1), 2.0 * np.max(new_2, 1)
    out = [lengths[:, 0], lengths[:, 1], lengths[:, 2]]
    out[a], out[b] = nd1, nd2
    return np.concatenate([new_centers, np.stack(out, axis=1)], axis=1)


def box_corners(center, size):
    """get_3d_box_batch (box_util.py:360-383) for heading 0 (obb[:, 6] = -0.0): (M,3),(M,3) -> (M,8,3) float64."""
    M = center.shape[0]
    heading = np.zeros(M) * -1
    R = np.zeros((M, 3, 3))
    c, s = np.cos(heading), np.sin(heading)
    R[:, 0, 0], R[:, 0, 2], R[:, 1, 1], R[:, 2, 0], R[:, 2, 2] = c, s, 1, -s, c
    l, w, h = size[:, 0:1], size[:, 1:2], size[:, 2:3]
    cr = np.zeros((M, 8, 3))
    cr[:, :, 0] = np.concatenate((l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2), -1)
    cr[:, :, 1] = np.concatenate((w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2), -1)
    cr[:, :, 2] = np.concatenate((h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2), -1)
    cr = np.matmul(cr, np.transpose(R, (0, 2, 1)))
    return cr + center[:, None, :]


class SceneStoreRef:
    """The scene cache of the reference (``self.scene_data``, lib/dataset.py:193-212) with its mutable relation labels."""

    def __init__(self, mean_size_arr, nyu40id2class, raw2label):
        self.scenes = {}
        self.mean_size_arr = np.asarray(mean_size_arr)
        self.nyu40id2class = dict(nyu40id2class)
        self.raw2label = dict(raw2label)

    def add_scene(self, scene_id, vert, ins, sem, bbox, x=None, y=None, z=None, multiview=None):
        self.scenes[scene_id] = dict(vert=np.array(vert), ins=np.array(ins), sem=np.array(sem), bbox=np.array(bbox),
                                     x=None if x is None else np.array(x), y=None if y is None else np.array(y),
                                     z=None if z is None else np.array(z),
                                     multiview=None if multiview is None else np.array(multiview))

    def get_item(self, scene_id, object_id, object_name, draws, num_points, use_height=True, use_normal=False,
                 augment=True, use_relation=True, use_color=False, use_multiview=False):
        sc = self.scenes[scene_id]
        vert, bbox = sc["vert"], sc["bbox"]
        if not use_color:
            pc = vert[:, 0:3]
            color = vert[:, 3:6]
        else:
            # lib/dataset.py:312-315: `point_cloud = mesh_vertices[:, 0:6]` is a VIEW of the cached scene, so the
            # normalisation below is written back into the cache: a scene's colours are re-normalised once more on
            # every access (float64 arithmetic, stored as the cache's float32)
            pc = vert[:, 0:6]
            pc[:, 3:6] = (pc[:, 3:6] - MEAN_COLOR_RGB) / 256.0
            color = pc[:, 3:6]
        if use_normal:
            pc = np.concatenate([pc, vert[:, 6:9]], 1)
        if use_multiview:                                                      # :321-328 (hdf5 rows of the scene)
            pc = np.concatenate([pc, sc["multiview"]], 1)
        if use_height:
            floor = np.percentile(pc[:, 2], 0.99)                              # :331
            pc = np.concatenate([pc, np.expand_dims(pc[:, 2] - floor, 1)], 1)
        ch = draws["choices"]
        pc = pc[ch]
        ins, sem, color = sc["ins"][ch], sc["sem"][ch], color[ch]

        tb = np.zeros((MAX_NUM_OBJ, 6))
        mask = np.zeros(MAX_NUM_OBJ)
        nb = min(bbox.shape[0], MAX_NUM_OBJ)
        mask[:nb] = 1
        tb[:nb] = bbox[:MAX_NUM_OBJ, 0:6]
        if augment:
            if draws["flip_x"]:
                pc[:, 0] = -1 * pc[:, 0]
                tb[:, 0] = -1 * tb[:, 0]
                if use_relation:
                    _swap02(sc["x"])
            if draws["flip_y"]:
                pc[:, 1] = -1 * pc[:, 1]
                tb[:, 1] = -1 * tb[:, 1]
                if use_relation:
                    _swap02(sc["y"])
            for ang, rot, ax in zip(draws["angles"], (rotx, roty, rotz), "xyz"):
                R = rot(ang)
                pc[:, 0:3] = np.dot(pc[:, 0:3], np.transpose(R))
                tb = rotate_aligned_boxes_along_axis(tb, R, ax)
            factor = list(draws["translation"])
            coords = pc[:, :3]
            coords += factor
            pc[:, :3] = coords
            tb[:, :3] += factor
        out = {}
        if use_relation:
            for a in "xyz":
                rel = np.zeros((MAX_NUM_OBJ, MAX_NUM_OBJ))
                rel[:nb, :nb] = sc[a]
                out[f"{a}_label"] = rel.astype(np.int64)
        votes = np.zeros([num_points, 3])
        vmask = np.zeros(num_points)
        for i in np.unique(ins):                                               # :415-427
            ind = np.where(ins == i)[0]
            if sem[ind[0]] in NYU40IDS:
                x = pc[ind, :3]
                center = 0.5 * (x.min(0) + x.max(0))
                votes[ind, :] = center - x
                vmask[ind] = 1.0
        votes = np.tile(votes, (1, 3))
        cls = [self.nyu40id2class[int(v)] for v in bbox[:nb, -2]]
        size_classes = np.zeros(MAX_NUM_OBJ)
        size_res = np.zeros((MAX_NUM_OBJ, 3))
        size_classes[:nb] = cls
        size_res[:nb] = tb[:nb, 3:6] - self.mean_size_arr[cls, :]
        ref_box = np.zeros(MAX_NUM_OBJ)
        ref_center, ref_cls, ref_res, ref_corners = np.zeros(3), 0, np.zeros(3), np.zeros((8, 3))
        for i, gt in enumerate(bbox[:nb, -1]):
            if gt == object_id:
                ref_box[i] = 1
                ref_center, ref_cls, ref_res = tb[i, 0:3], size_classes[i], size_res[i]
                size = self.mean_size_arr[int(ref_cls), :] + ref_res
                ref_corners = box_corners(ref_center[None], size[None])[0]
        gt_corners = np.zeros((MAX_NUM_OBJ, 8, 3))
        gt_corners[:nb] = box_corners(tb[:nb, 0:3], self.mean_size_arr[size_classes[:nb].astype(np.int64), :] + size_res[:nb])
        ids = np.zeros(MAX_NUM_OBJ)
        ids[:nb] = bbox[:, -1][:nb]
        sem_cls = np.zeros(MAX_NUM_OBJ)
        sem_cls[:nb] = cls
        gmask = np.zeros(MAX_NUM_OBJ)
        gmask[:nb] = 1
        out.update({
            "point_clouds": pc.astype(np.float32), "pcl_color": color,
            "center_label": tb.astype(np.float32)[:, 0:3],
            "heading_class_label": np.zeros(MAX_NUM_OBJ, np.int64), "heading_residual_label": np.zeros(MAX_NUM_OBJ, np.float32),
            "size_class_label": size_classes.astype(np.int64), "size_residual_label": size_res.astype(np.float32),
            "num_bbox": np.array(nb).astype(np.int64), "sem_cls_label": sem_cls.astype(np.int64),
            "scene_object_ids": ids.astype(np.int64), "box_label_mask": mask.astype(np.float32),
            "box_label_mask_int": mask.astype(np.int64), "vote_label": votes.astype(np.float32),
            "vote_label_mask": vmask.astype(np.int64), "ref_box_label": ref_box.astype(np.int64),
            "ref_center_label": ref_center.astype(np.float32),
            "ref_heading_class_label": np.array(0).astype(np.int64), "ref_heading_residual_label": np.array(0).astype(np.int64),
            "ref_size_class_label": np.array(int(ref_cls)).astype(np.int64),
            "ref_size_residual_label": ref_res.astype(np.float32), "ref_box_corner_label": ref_corners.astype(np.float64),
            "gt_box_corner_label": gt_corners.astype(np.float64), "gt_box_masks": gmask.astype(np.int64),
            "gt_box_object_ids": ids.astype(np.int64), "object_id": np.array(int(object_id)).astype(np.int64),
            "object_cat": np.array(self.raw2label.get(object_name, 17)).astype(np.int64),
        })
        return out


def _swap02(lab):
    zero, two = np.where(lab == 0), np.where(lab == 2)
    lab[zero] = 2
    lab[two] = 0


def language_arrays(tokens, glove, word2idx):
    """``_tranform_des`` (lib/dataset.py:76-118) for one description + the fields __getitem__ derives (:299-302,
    :477-481): returns lang_feat (32,300) f32, lang_ids (32,) i64, lang_label (33,) i64, lang_len."""
    toks = ["sos"] + list(tokens)[:MAX_DES_LEN] + ["eos"]
    emb = np.zeros((MAX_DES_LEN + 2, 300))
    lab = np.zeros(MAX_DES_LEN + 2)
    for i, t in enumerate(toks):
        try:
            emb[i] = glove[t]
            lab[i] = word2idx[t]
        except KeyError:
            emb[i] = glove["unk"]
            lab[i] = word2idx["unk"]
    n = len(tokens) + 2
    n = n if n <= MAX_DES_LEN + 2 else MAX_DES_LEN + 2
    return (emb.astype(np.float32), lab.astype(np.int64),
            np.concatenate((np.array([1]), lab), axis=0).astype(np.int64), np.array(n).astype(np.int64))

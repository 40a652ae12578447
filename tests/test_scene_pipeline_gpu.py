"""SURVEY.md section 8(f) rank 4 -- the input pipeline, GPU leg: spacap3d_amd.dataset.DeviceSceneDataset (HBM-resident
scenes, subsample / augmentation / votes in HIP kernels through the C ABI, labels in batched torch ops) against the
reference's own outputs (tests/golden/scene_pipeline.npz) and the numpy restatement, fed with the reference's random
draws.  Bar: every integer / mask / label array bit-exact; coordinates and votes within 1 float32 ulp (the device
evaluates the float64 rotation products without FMA, numpy's BLAS may fuse them before the float32 rounding)."""
import numpy as np
import pytest
import torch

from oracle import scene_pipeline_ref as R
from tests.test_scene_pipeline import load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EXACT_F = {"pcl_color", "box_label_mask"}


def _device_dataset(fx, num_points):
    from spacap3d_amd.dataset import DeviceSceneDataset
    ds = DeviceSceneDataset(DEV, fx["mean_size_arr"], dict(zip(fx["nyu40id2class_keys"].tolist(), fx["nyu40id2class_vals"].tolist())),
                            dict(zip(fx["raw2label_names"].tolist(), fx["raw2label_vals"].tolist())), num_points=num_points,
                            max_instances=64)
    for sid in fx["scene_ids"].tolist():
        ds.add_scene(sid, fx[f"{sid}/vert"], fx[f"{sid}/ins"], fx[f"{sid}/sem"], fx[f"{sid}/bbox"], fx[f"{sid}/x"],
                     fx[f"{sid}/y"], fx[f"{sid}/z"])
    glove = dict(zip(fx["glove_words"].tolist(), fx["glove_vecs"]))
    w2i = dict(zip(fx["vocab_words"].tolist(), fx["vocab_ids"].tolist()))
    for i in range(int(fx["n_items"])):
        feat, ids, _, n = R.language_arrays(str(fx["item_tokens"][i]).split("|"), glove, w2i)
        ds.add_item(str(fx["item_scene"][i]), int(fx["item_object"][i]), str(fx["item_object_name"][i]), feat, ids, int(n),
                    ann_id=int(fx["item_ann"][i]))
    return ds


def _compare(name, got, want):
    got = got.cpu().numpy()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert got.dtype == want.dtype, (name, got.dtype, want.dtype)
    if want.dtype.kind in "iu" or name in EXACT_F:
        assert np.array_equal(got, want), name
    elif want.dtype == np.float32:
        ulp = np.spacing(np.maximum(np.abs(want), np.float32(1e-3)))
        bad = np.abs(got.astype(np.float64) - want) > 1.01 * ulp if name in ("point_clouds", "vote_label") else \
            np.abs(got.astype(np.float64) - want) > 2e-6 * (1 + np.abs(want))
        assert not bad.any(), (name, int(bad.sum()), float(np.abs(got - want).max()))
    else:
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (name, float(np.abs(got - want).max()))


def test_device_pipeline_reproduces_the_reference_items_in_order():
    fx, _ = load_fixture()
    P = int(fx["num_points"])
    ds = _device_dataset(fx, P)
    exact_points = total_points = 0
    for step, idx in enumerate(fx["order"].tolist()):
        sid = str(fx["item_scene"][idx])
        draws = [R.draws_from_seed(int(fx[f"step{step}/seed"]), fx[f"{sid}/vert"].shape[0], P)]
        d = ds.batch([idx], draws)
        for k in [k[len(f"step{step}/"):] for k in fx.files if k.startswith(f"step{step}/")]:
            if k == "seed":
                continue
            assert k in d, k
            _compare(k, d[k][0], fx[f"step{step}/{k}"])
        g = d["point_clouds"][0].cpu().numpy()
        exact_points += int((g == fx[f"step{step}/point_clouds"]).all(1).sum())
        total_points += P
    assert exact_points >= 0.999 * total_points   # the 1-ulp allowance is the rare exception, not the rule


def test_a_mixed_batch_equals_the_items_processed_one_by_one():
    """B = 4 descriptions over both scenes (one scene twice: its relation-label flip state advances inside the batch)."""
    fx, store = load_fixture()
    P = int(fx["num_points"])
    ds = _device_dataset(fx, P)
    idxs = [0, 4, 2, 5]
    draws = [R.draws_from_seed(50 + i, fx[f"{str(fx['item_scene'][i])}/vert"].shape[0], P) for i in idxs]
    d = ds.batch(idxs, draws)
    assert d["point_clouds"].shape == (4, P, 4) and d["vote_label"].shape == (4, P, 9)
    for b, i in enumerate(idxs):
        want = store.get_item(str(fx["item_scene"][i]), int(fx["item_object"][i]), str(fx["item_object_name"][i]), draws[b], P)
        for k, w in want.items():
            _compare(k, d[k][b], w)


def test_own_draws_feed_a_training_step():
    """draw() -> batch() -> the model: keys, dtypes and shapes are what the training step consumes."""
    from spacap3d_amd.engine import Trainer
    from spacap3d_amd.spacapnet import build_default
    fx, _ = load_fixture()
    ds = _device_dataset(fx, 2048)
    g = torch.Generator(device=DEV).manual_seed(0)
    d = ds.batch([0, 3], ds.draw([0, 3], generator=g))
    ch, aug = ds.draw([0, 3, 1], generator=g)
    n = torch.tensor([fx[f"{str(fx['item_scene'][i])}/vert"].shape[0] for i in (0, 3, 1)], device=DEV)
    assert ch.shape == (3, 2048) and bool((ch >= 0).all()) and bool((ch < n[:, None]).all())
    assert len(torch.unique(ch[0])) == 2048                       # 5000 vertices >= 2048 points: no repeats
    assert bool((aug[:, 2] == 1).all()) and bool((aug[:, 29:32].abs() <= 0.5).all())
    assert d["point_clouds"].dtype == torch.float32 and d["vote_label_mask"].dtype == torch.int64
    assert 0 < int(d["vote_label_mask"].sum()) < 2 * 2048
    torch.manual_seed(0)
    vocab = int(fx["vocab_ids"].max()) + 1
    model = build_default(vocab_size=vocab, num_proposal=32, N=1, d_ff=64, mean_size_arr=fx["mean_size_arr"]).to(DEV).train()
    tr = Trainer(model, fx["mean_size_arr"])
    loss = tr.step(d)
    assert torch.isfinite(loss)


def _device_dataset_feats(fx, num_points, color_renorm="per_access", **kw):
    from spacap3d_amd.dataset import DeviceSceneDataset
    from tests.test_scene_pipeline import multiview_rows
    ds = DeviceSceneDataset(DEV, fx["mean_size_arr"], dict(zip(fx["nyu40id2class_keys"].tolist(), fx["nyu40id2class_vals"].tolist())),
                            dict(zip(fx["raw2label_names"].tolist(), fx["raw2label_vals"].tolist())), num_points=num_points,
                            max_instances=64, color_renorm=color_renorm, **kw)   # (the fixture holds the reference's items, quirk included)
    for sid in fx["scene_ids"].tolist():
        v = fx[f"{sid}/vert"]
        ds.add_scene(sid, v, fx[f"{sid}/ins"], fx[f"{sid}/sem"], fx[f"{sid}/bbox"], fx[f"{sid}/x"], fx[f"{sid}/y"], fx[f"{sid}/z"],
                     multiview=multiview_rows(sid, v.shape[0]) if kw.get("use_multiview") else None)
    for i in range(int(fx["n_items"])):
        ds.add_item(str(fx["item_scene"][i]), int(fx["item_object"][i]), str(fx["item_object_name"][i]), ann_id=int(fx["item_ann"][i]))
    return ds


@pytest.mark.parametrize("tag", ["color_normal", "multiview_normal", "all"])
def test_device_pipeline_colour_and_multiview_match_the_reference(tag):
    """BASELINE configs 3 / 4 input channels against what the reference's own dataset class returned (item by item, in
    order: scene 0 is visited four times, each time with its colours normalised once more)."""
    import os
    from tests.test_scene_pipeline import FEATS, G
    fx, _ = load_fixture()
    ff = np.load(os.path.join(G, "scene_pipeline_feats.npz"))
    ds = _device_dataset_feats(fx, 1024, **FEATS[tag])
    for step in range(5):
        idx = int(ff[f"{tag}/step{step}/idx"])
        sid = str(fx["item_scene"][idx])
        draws = [R.draws_from_seed(int(ff[f"{tag}/step{step}/seed"]), fx[f"{sid}/vert"].shape[0], 1024)]
        d = ds.batch([idx], draws)
        want = ff[f"{tag}/step{step}/point_clouds_rows8"]
        got = d["point_clouds"][0].cpu().numpy()[::8]
        assert got.shape == want.shape and got.dtype == want.dtype
        assert np.array_equal(got[:, 3:], want[:, 3:]), (tag, step)          # every feature channel: bit-exact
        ulp = np.spacing(np.maximum(np.abs(want[:, :3]), np.float32(1e-3)))
        assert not (np.abs(got[:, :3].astype(np.float64) - want[:, :3]) > 1.01 * ulp).any()
        assert np.array_equal(d["pcl_color"][0].cpu().numpy(), ff[f"{tag}/step{step}/pcl_color"]), (tag, step)
        assert np.array_equal(d["vote_label_mask"][0].cpu().numpy(), ff[f"{tag}/step{step}/vote_label_mask"])


def test_a_batch_that_repeats_a_scene_sees_successive_colour_states():
    """Two descriptions of scene 0 and one of scene 1 in ONE batch: the second visit of scene 0 must see the colours
    normalised twice (the reference's cache semantics), exactly as processing the items one by one."""
    from tests.test_scene_pipeline import FEATS, multiview_rows
    fx, _ = load_fixture()
    kw = FEATS["all"]
    store = R.SceneStoreRef(fx["mean_size_arr"], dict(zip(fx["nyu40id2class_keys"].tolist(), fx["nyu40id2class_vals"].tolist())),
                            dict(zip(fx["raw2label_names"].tolist(), fx["raw2label_vals"].tolist())))
    for sid in fx["scene_ids"].tolist():
        v = fx[f"{sid}/vert"]
        store.add_scene(sid, v, fx[f"{sid}/ins"], fx[f"{sid}/sem"], fx[f"{sid}/bbox"], fx[f"{sid}/x"], fx[f"{sid}/y"],
                        fx[f"{sid}/z"], multiview=multiview_rows(sid, v.shape[0]))
    ds = _device_dataset_feats(fx, 1024, **kw)
    idxs = [0, 3, 1]        # items 0 and 1 describe scene 0, item 3 scene 1
    draws = [R.draws_from_seed(70 + i, fx[f"{str(fx['item_scene'][i])}/vert"].shape[0], 1024) for i in idxs]
    d = ds.batch(idxs, draws)
    assert d["point_clouds"].shape == (3, 1024, 138)
    for b, i in enumerate(idxs):
        want = store.get_item(str(fx["item_scene"][i]), int(fx["item_object"][i]), str(fx["item_object_name"][i]), draws[b], 1024, **kw)
        got = d["point_clouds"][b].cpu().numpy()
        assert np.array_equal(got[:, 3:], want["point_clouds"][:, 3:]), (b, i)
        assert np.array_equal(d["pcl_color"][b].cpu().numpy(), want["pcl_color"]), (b, i)
    a, c = d["pcl_color"][0].std(0).max(), d["pcl_color"][2].std(0).max()
    assert float(a) > 100 * float(c) > 0


def test_colours_normalised_once_by_default_and_resettable_in_the_parity_mode():
    """with ``color_renorm="once"`` two visits of a scene return the same colours, (raw - mean) / 256; the
    parity mode normalises once more per visit and ``reset_colors()`` restores the loaded state."""
    fx, _ = load_fixture()
    kw = dict(use_color=True, use_normal=True, use_multiview=False)
    once = _device_dataset_feats(fx, 2000, color_renorm="once", **kw)      # the default mode keeps no raw copy
    assert all("color0" not in sc for sc in once._scenes)
    g = torch.Generator(device=DEV).manual_seed(1)
    draws = once.draw([0], g)
    a = once.batch([0], draws=draws)["point_clouds"][..., 3:6].clone()
    b = once.batch([0], draws=draws)["point_clouds"][..., 3:6]
    assert torch.equal(a, b) and float(a.abs().max()) <= 1.0
    with pytest.raises(AttributeError):                # the mode is fixed at construction (which copies a scene keeps depends on it)
        once.color_renorm = "per_access"
    once.reset_colors()                                # a no-op in this mode, never a KeyError
    ds = _device_dataset_feats(fx, 2000, **kw)         # the parity mode
    first = ds.batch([0], draws=draws)["point_clouds"][..., 3:6].clone()
    assert torch.equal(first, a)                       # first visit: (raw - mean) / 256, what "once" serves every time
    c = ds.batch([0], draws=draws)["point_clouds"][..., 3:6].clone()
    assert not torch.equal(a, c)                       # normalised a second time
    ds.reset_colors()
    d = ds.batch([0], draws=draws)["point_clouds"][..., 3:6]
    assert torch.equal(d, a)                           # loaded colours, normalised once by this visit


"""GPU: the data path and evaluation harness against their restatement in torch CPU ops
(oracle.prepare_clip / oracle.eval_metrics, citing datasets/loader.py and main.py:484-543)."""
import numpy as np
import pytest
import torch

from oracle import glfusion_ref as orc   # the checker (tests only)

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("view", ["1", "2", "3", "4"])
@pytest.mark.parametrize("h0,w0,t,offset", [(200, 160, 7, None), (600, 800, 3, (5, 31)), (97, 131, 1, (32, 0)), (144, 144, 4, (0, 32))])
def test_prepare_frames_matches_loader_chain(view, h0, w0, t, offset):
    """Resize(nearest, 144) -> crop 112 -> part masks -> 5 class channels -> / 255 -> frame reshape: bit-exact for the
    masks (index work), exact for the frames (one fp32 multiply), on ragged volume sizes, up- and down-sampling."""
    from glfusion_amd import data
    g = torch.Generator().manual_seed(h0 * 7 + t)
    img = (torch.rand(h0, w0, t, generator=g) * 255).floor()
    lab = (torch.rand(h0, w0, t, generator=g) * 5).floor()           # ids 0..4 (ids beyond the view's parts are dropped)
    want_f, want_m = orc.prepare_clip(img, lab, view, offset)
    got_f, got_m = data.prepare_frames(img.to(DEV), lab.to(DEV), view, train=False, crop_offset=offset)
    assert tuple(got_f.shape) == (t, 1, 112, 112) and tuple(got_m.shape) == (t, 5, 112, 112)
    assert torch.equal(got_m.cpu(), want_m)
    assert torch.equal(got_f.cpu(), want_f)
    # unlabelled clips keep raw grey levels (loader.py:325), a missing label volume gives no masks
    raw, none = data.prepare_frames(img.to(DEV), None, view, crop_offset=offset, labelled=False)
    assert none is None and torch.equal(raw.cpu(), orc.prepare_clip(img, lab, view, offset, labelled=False)[0])


def test_prepare_frames_train_crop_and_errors():
    from glfusion_amd import data
    img = (torch.rand(50, 60, 2) * 255).floor().to(DEV)
    lab = torch.zeros(50, 60, 2, device=DEV)
    torch.manual_seed(3)
    f, m = data.prepare_frames(img, lab, "1", train=True)
    torch.manual_seed(3)
    oy = int(torch.randint(0, 33, ()).item()); ox = int(torch.randint(0, 33, ()).item())
    assert torch.equal(f.cpu(), orc.prepare_clip(img.cpu(), lab.cpu(), "1", (oy, ox))[0])
    assert float(m.sum()) == 0.0
    with pytest.raises(RuntimeError, match="crop window"):
        data.prepare_frames(img, lab, "1", crop_offset=(40, 0))
    with pytest.raises(KeyError):
        data.prepare_frames(img, lab, "9")


def test_part_overlap_counts_bit_exact():
    from glfusion_amd import data, ops
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(6, 5, 112, 112, generator=g) * 3
    tgt = (torch.rand(6, 5, 112, 112, generator=g) < 0.3).float()
    counts = data.part_overlap_counts(logits.to(DEV), tgt.to(DEV)).cpu()
    pred = orc.binarize(logits)
    for c in range(5):
        o, t = pred[:, c].reshape(-1).double(), tgt[:, c].reshape(-1).double()
        want = [int((o * t).sum()), int((o * (1 - t)).sum()), int(((1 - o) * t).sum()), int(((1 - o) * (1 - t)).sum())]
        assert counts[c].tolist() == want, c
    assert counts.sum(0).tolist() == ops.overlap_counts(logits.to(DEV), tgt.to(DEV)).cpu().tolist()


def test_eval_harness_vs_oracle():
    """Trainer.eval on synthetic patient volumes against the oracle doing what main.py:484-543 does: frames of every patient
    through the (oracle) model, predictions and masks concatenated, overlap metrics of the view and Dice per part."""
    from glfusion_amd.data import SyntheticPatients
    from glfusion_amd.engine import Trainer
    views = ["1", "4"]
    cfg = {"train": {"batch_size": 1, "num_epochs": 1, "clip_length": 3, "view_num": views, "test_view": views, "save_dir": "/tmp/glf_eval",
                     "iters_per_epoch": 1, "global_rank": 0},
           "net": {"opt": {"opt_name": "Adam", "lr": 3e-4, "params": (0.9, 0.999), "weight_decay": 1e-5}}}
    t = Trainer(cfg)
    ref = orc.Global_and_Local(views)
    orc.closed_form_fill(ref, salt=9)
    ref.eval()
    t.model.load_state_dict(ref.state_dict(), strict=True)
    patients = SyntheticPatients(views, 2, clip_length=3, h0=150, w0=170, device=DEV, seed=5)
    got = t.eval(patients=patients)
    preds, masks = {v: [] for v in views}, {v: [] for v in views}
    with torch.no_grad():
        for sample in patients:
            imgs, mk = {}, {}
            for v in views:
                imgs[v], mk[v] = orc.prepare_clip(sample[v][0].cpu(), sample[v][1].cpu(), v)
            out = ref(imgs)[0]
            for v in views:
                preds[v].append(out[v])
                masks[v].append(mk[v])
    for v in views:
        whole, parts = orc.eval_metrics(torch.cat(preds[v]), torch.cat(masks[v]))
        assert np.allclose(got[v], whole, atol=1e-4, rtol=0), (v, got[v], whole)            # north_star: Dice within 1e-4
        assert np.allclose(t.eval_report["part_dice"][v], parts, atol=1e-4, rtol=0), (v, t.eval_report["part_dice"][v], parts)
        assert max(parts) > 0.0                                                              # the fixture is not degenerate


def test_dataset_shim_and_per_epoch_validation(tmp_path):
    """SegPAHDataset items come out in the reference's layouts ([1,112,112(,T)] / [5,112,112(,T)], images / 255), and
    Trainer.validation_and_test (main.py:279-415) scores the Inner-val / Inner-test splits: metrics in [0, 1], five part Dice
    values per view, the validation Dice as return value; with a checkpoint directory it scores every net_%05d.pth."""
    import random
    from glfusion_amd import data
    from glfusion_amd.engine import Trainer
    infos = data.synthetic_infos(["4"], 4, clip_length=6, device="cpu", seed=2)
    random.seed(0)
    tr = data.SegPAHDataset(infos, is_train=True, data_list=list(infos), view_num=["4"], single_frame=True, device=DEV, crop_seed=1)
    img, mask, idx = tr[5]
    assert tuple(img.shape) == (1, 112, 112) and tuple(mask.shape) == (5, 112, 112) and float(img.max()) <= 1.0
    assert set(mask.unique().tolist()) <= {0.0, 1.0} and float(mask[4].sum()) == 0.0            # view 4 never fills channel 4 (PA)
    ev = data.SegPAHDataset(infos, is_train=False, data_list=["0_1"], view_num=["4"], single_frame=False, clip_length=5, device=DEV)
    cimg, cmask, _ = ev[0]
    assert tuple(cimg.shape[:3]) == (1, 112, 112) and tuple(cmask.shape[:3]) == (5, 112, 112) and cimg.shape[-1] == cmask.shape[-1] <= 4
    config = {"train": {"view_num": ["4"], "test_view": ["4"], "num_epochs": 1, "batch_size": 2, "iters_per_epoch": 1, "clip_length": 6,
                        "save_dir": str(tmp_path), "validate_every_epoch": False},
              "net": {"opt": {"opt_name": "Adam", "lr": 3e-4, "weight_decay": 1e-5}}}
    t = Trainer(config)
    random.seed(3)
    val = t.validation_and_test(net_root=None, infos=infos, val_list=("0_0", "0_2"), test_list=("0_1", "0_3"))
    rep = t.validation_report
    assert 0.0 <= val <= 1.0 and set(rep) == {"Inner-val", "Inner-test", "val_dice"}
    for split in ("Inner-val", "Inner-test"):
        m = rep[split]["4"]
        assert all(0.0 <= x <= 1.0 for x in m["metrics"]) and len(m["part_dice"]) == 5 and m["loss"] > 0
    t.save(0); t.save(1)
    assert open(tmp_path / "latest.ckpt").read() == "00001\n"
    best, dices = t.validation_and_test(net_root=str(tmp_path), infos=infos, val_list=("0_0", "0_2"), test_list=("0_1",), first_scored=0)
    assert len(dices) == 2 and best in (0, 1)


def test_dataset_reads_nifti_paths_like_the_reference(tmp_path):
    """The reference's infos hold NIfTI paths (loader.py:233-234 loads them with nibabel).  The same patients written to
    .nii.gz (uint8 images, uint8 label maps, W x H x T) and handed over as paths give bit-identical items to the in-memory
    volumes, in train (random frame + random crop) and eval (clip) mode."""
    import random
    import numpy as np
    from glfusion_amd import data, nifti
    infos = data.synthetic_infos(["4"], 3, clip_length=6, device="cpu", seed=4)
    on_disk = {}
    for pid, e in infos.items():
        ent = {"dataset_name": e["dataset_name"], "fold": e["fold"], "views_images": {}, "views_labels": {}}
        for v in e["views_images"]:
            img, lab = e["views_images"][v](), e["views_labels"][v]()
            img, lab = np.asarray(img.cpu()), np.asarray(lab.cpu())
            assert float(np.abs(img - np.round(img)).max()) == 0.0 and img.min() >= 0 and img.max() <= 255     # 8-bit echo frames
            pi, pl = tmp_path / f"{pid}_{v}_img.nii.gz", tmp_path / f"{pid}_{v}_lab.nii.gz"
            nifti.write(pi, img.astype(np.uint8)); nifti.write(pl, lab.astype(np.uint8))
            ent["views_images"][v], ent["views_labels"][v] = str(pi), pl                                   # str and PathLike
        on_disk[pid] = ent
    for kw in (dict(is_train=True, single_frame=True, crop_seed=7), dict(is_train=False, single_frame=False, clip_length=5)):
        outs = []
        for src in (infos, on_disk):
            random.seed(11)
            ds = data.SegPAHDataset(src, data_list=list(src), view_num=["4"], device=DEV, **kw)
            outs.append([ds[i] for i in range(len(ds))])
        for a, b in zip(*outs):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]

"""Pin the CPU oracle (oracle/manipose_ref.py) against golden vectors produced by the reference itself
(oracle/gen_golden.py) and against the closed-form known answers of SURVEY.md section 4."""
import numpy as np
import pytest
import torch

import manipose_ref as orc
from helpers import GOLDEN, fixture_masks, fixture_state, h36m_calibration, load_fixture, raw_dataset_dicts

TOL = dict(rtol=1e-5, atol=2e-6)


def _run_rmcl(fx, masks=None):
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    X, y = torch.from_numpy(fx["X"]), torch.from_numpy(fx["y"])
    poses, scores = orc.rmcl_manifold_forward(X, st, orc.oracle_cfg(fx["cfg"]), masks)
    total, terms = orc.rmcl_training_loss(poses, scores, y)
    total.backward()
    return st, poses, scores, total, terms


@pytest.mark.parametrize("name", ["rmcl_tiny", "rmcl_small"])
def test_rmcl_forward_loss_grads(name):
    fx = load_fixture(name)
    st, poses, scores, total, terms = _run_rmcl(fx)
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], **TOL)
    np.testing.assert_allclose(scores.detach().numpy(), fx["scores"], **TOL)
    got = np.array([terms[k].item() for k in ("wloss", "score_reg", "vloss", "sreg")])
    np.testing.assert_allclose(got, fx["loss_terms"], rtol=1e-5)
    np.testing.assert_allclose(total.item(), float(fx["loss_total"]), rtol=1e-5)
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_rmcl_intermediates_and_aggregation():
    fx = load_fixture("rmcl_small")
    st = fixture_state(fx)
    cfg = fx["cfg"]
    X, y = torch.from_numpy(fx["X"]), torch.from_numpy(fx["y"])
    rot, _ = orc.rmcl_rot_forward(X, st, "rotations_module.", cfg["depth_rot"], cfg["heads_rot"], cfg["n_hyp"])
    bl = orc.bones_forward(X, st, "segments_module.", cfg["depth_seg"], cfg["heads_seg"], cfg["num_bones"])
    np.testing.assert_allclose(rot.numpy(), fx["rot6d"], **TOL)
    np.testing.assert_allclose(bl.numpy(), fx["bones"], **TOL)
    poses, scores = torch.from_numpy(fx["poses"]), torch.from_numpy(fx["scores"])
    val, idx = orc.wta_l2_loss_and_activate_head(poses, y, torch.tensor(orc.STANDARD_H36M_WEIGHTS))
    np.testing.assert_array_equal(idx.numpy(), fx["wta_idx"])
    np.testing.assert_allclose(val.numpy(), fx["wta_val"], **TOL)
    np.testing.assert_allclose(orc.aggregate(poses, scores, "weighted_ave").numpy(), fx["agg_weighted"], **TOL)
    np.testing.assert_allclose(orc.aggregate(poses, scores, "best_score").numpy(), fx["agg_best"], **TOL)
    e, p = orc.aggregate(poses, mode="oracle", ground_truth=y)
    np.testing.assert_allclose(p.numpy(), fx["agg_oracle"], **TOL)
    np.testing.assert_allclose(e.numpy(), fx["agg_oracle_err"], **TOL)
    np.testing.assert_allclose(orc.mpjpe_error(orc.aggregate(poses, scores), y).item(),
                               float(fx["mpjpe_weighted"]), rtol=1e-6)
    with pytest.raises(ValueError):
        orc.aggregate(poses, scores, mode="median")


def test_manifold_single_hypothesis():
    fx = load_fixture("manifold_k1")
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    X, y = torch.from_numpy(fx["X"]), torch.from_numpy(fx["y"])
    pred = orc.manifold_forward(X, st, orc.oracle_cfg(fx["cfg"]))
    total, terms = orc.manifold_training_loss(pred, y)
    total.backward()
    np.testing.assert_allclose(pred.detach().numpy(), fx["poses"], **TOL)
    got = np.array([terms[k].item() for k in ("wloss", "vloss", "sreg")])
    np.testing.assert_allclose(got, fx["loss_terms"], rtol=1e-5)
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_droppath_mask_placement():
    """Train mode with the masks the (stand-in) DropPath drew injected into the oracle: pins WHERE the
    reference applies stochastic depth (mix_ste.py:352-358) and its per-dim-0-row granularity."""
    fx = load_fixture("rmcl_tiny_droppath")
    masks = fixture_masks(fx)
    assert len(masks) == 4
    st, poses, scores, total, _ = _run_rmcl(fx, masks)
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], **TOL)
    np.testing.assert_allclose(total.item(), float(fx["loss_total"]), rtol=1e-5)
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_decoder_with_degenerate_rotations():
    fx = load_fixture("decoder")
    rot = torch.from_numpy(fx["rot6d"]).requires_grad_(True)
    bl = torch.from_numpy(fx["bones"]).requires_grad_(True)
    poses = orc.pose_decoder(rot, bl)
    (poses * torch.from_numpy(fx["gpos"])).sum().backward()
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], **TOL)
    np.testing.assert_allclose(rot.grad.numpy(), fx["g_rot6d"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bl.grad.numpy(), fx["g_bones"], rtol=1e-4, atol=1e-5)
    assert np.all(poses.detach().numpy()[:, 0] == 0)          # root joint exactly 0


def test_known_answer_tpose():
    """identity 6-D input -> T-pose with the stated joint coordinates."""
    fx = load_fixture("decoder")
    ident = torch.tensor([1., 0, 0, 0, 1, 0]).repeat(1, 17, 1)
    tp = orc.pose_decoder(ident, torch.from_numpy(fx["tpose_lens"]))[0]
    np.testing.assert_allclose(tp.numpy(), fx["tpose"][0], atol=1e-7)
    expect = {1: (.2, 0, 0), 2: (.2, -.5, 0), 3: (.2, -1, 0), 4: (-.2, 0, 0), 10: (0, .8, 0),
              13: (-1, .4, 0), 16: (1, .4, 0)}
    for j, xyz in expect.items():
        np.testing.assert_allclose(tp[j].numpy(), np.array(xyz, dtype=np.float32), atol=1e-6)


def test_manifold_property_segment_lengths():
    """Every decoded pose has exactly the predicted segment lengths."""
    fx = load_fixture("rmcl_small")
    poses = torch.from_numpy(fx["poses"])                      # (B,H,L,J,3)
    bones = torch.from_numpy(fx["bones"]).abs()[:, :, 0]       # (B,S)
    par = torch.tensor(orc.H36M_PARENTS[1:])
    seg = (poses[..., 1:, :] - poses[..., par, :]).norm(dim=-1)
    np.testing.assert_allclose(seg.numpy(), bones[:, None, None, :].expand_as(seg).numpy(), atol=2e-6)


def test_loss_terms_and_grads():
    fx = load_fixture("loss")
    poses = torch.from_numpy(fx["poses"]).requires_grad_(True)
    scores = torch.from_numpy(fx["scores"]).requires_grad_(True)
    total, terms = orc.rmcl_training_loss(poses, scores, torch.from_numpy(fx["y"]))
    total.backward()
    got = np.array([terms[k].item() for k in ("wloss", "score_reg", "vloss", "sreg")])
    np.testing.assert_allclose(got, fx["loss_terms"], rtol=1e-6)
    np.testing.assert_allclose(poses.grad.numpy(), fx["g_poses"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(scores.grad.numpy(), fx["g_scores"], rtol=1e-5, atol=1e-8)


def test_adam_step_restated():
    fx = load_fixture("rmcl_tiny")
    for k in [k[7:] for k in fx if k.startswith("adam1::")]:
        p = torch.from_numpy(fx["w::" + k])
        g = torch.from_numpy(fx["g::" + k])
        p1, _, _ = orc.adam_step(p, g, torch.zeros_like(p), torch.zeros_like(p), 1)
        np.testing.assert_allclose(p1.numpy(), fx["adam1::" + k], rtol=1e-6, atol=1e-9)


def test_param_counts_and_key_layout(golden_dir):
    z = np.load(golden_dir + "/param_counts.npz")
    assert int(z["rmcl_T243_K5"]) == 34440062 and int(z["rmcl_T81_K5"]) == 34336382
    st = orc.make_state(orc.FULL_CFG)
    assert sum(v.numel() for v in st.values()) == 34440062 and len(st) == 290
    want = open(golden_dir + "/state_dict_keys_T243_K5.txt").read().split()
    got = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in st.items())
    assert got == want


# ---------------------------------------------------------------------------------------------------------------------
# evaluation analytics: oracle restatement vs the reference's own metric functions (metrics.npz)
# ---------------------------------------------------------------------------------------------------------------------
def _metric_cases(fx):
    for nm in ("pred", "rigid"):
        x, gt = torch.from_numpy(fx[nm]), torch.from_numpy(fx["gt"])
        yield nm, x, gt, x.permute(0, 3, 2, 1), gt.permute(0, 3, 2, 1)


def test_oracle_consistency_metrics_match_reference():
    fx = load_fixture("metrics")
    for nm, x, gt, jc, gj in _metric_cases(fx):
        for mode in ("average", "sum", "std", "min", "max"):
            np.testing.assert_allclose(orc.segments_time_consistency(jc, mode).numpy(), fx[f"{nm}.stc.{mode}"], rtol=1e-6, atol=0)
        for mode in ("average", "sum", "std"):
            np.testing.assert_allclose(orc.segments_time_consistency_per_bone(jc, mode).numpy(), fx[f"{nm}.stc_per_bone.{mode}"],
                                       rtol=1e-6)
        flat = jc.permute(1, 2, 0, 3).reshape(1, 3, 17, -1)
        np.testing.assert_allclose(orc.segments_time_consistency(flat, "std").numpy(), fx[f"{nm}.stc_flat.std"], rtol=1e-6)
        for sq in (False, True):
            for mode in ("average", "sum"):
                np.testing.assert_allclose(orc.sagittal_symmetry(jc, mode, sq).numpy(), fx[f"{nm}.sym.{mode}.{int(sq)}"], rtol=1e-6)
                np.testing.assert_allclose(orc.sagittal_symmetry_per_bone(jc, mode, sq).numpy(),
                                           fx[f"{nm}.sym_per_bone.{mode}.{int(sq)}"], rtol=1e-6)
        for signed in (False, True):
            for mode in ("average", "sum"):
                np.testing.assert_allclose(orc.segments_len_err(jc, gj, mode, signed).numpy(),
                                           fx[f"{nm}.len_err.{mode}.{int(signed)}"], rtol=1e-6, atol=1e-9)


def test_oracle_error_metrics_and_pck_match_reference():
    fx = load_fixture("metrics")
    mask = torch.from_numpy(fx["mask"])
    for nm, x, gt, jc, gj in _metric_cases(fx):
        for mode in ("average", "sum"):
            np.testing.assert_allclose(orc.mpjpe_error(x, gt, mode).numpy(), fx[f"{nm}.mpjpe.{mode}"], rtol=1e-6)
            np.testing.assert_allclose(orc.mse_error(x, gt, mode).numpy(), fx[f"{nm}.mse.{mode}"], rtol=1e-6)
            np.testing.assert_allclose(orc.jointwise_error(x, gt, mode).numpy(), fx[f"{nm}.jw_err.{mode}"], rtol=1e-6)
            np.testing.assert_allclose(orc.jointwise_error(x, gt, mode, squared=True).numpy(), fx[f"{nm}.jw_mse.{mode}"], rtol=1e-6)
        np.testing.assert_allclose(orc.eval_velocity_error(x, gt, 1, False).numpy(), fx[f"{nm}.vel"], rtol=1e-6)
        np.testing.assert_allclose(orc.eval_velocity_error(x, gt, 1, True).numpy(), fx[f"{nm}.vel_sq"], rtol=1e-6)
        xf, gf = x.reshape(-1, 17, 3), gt.reshape(-1, 17, 3)
        for al in ("none", "scale", "procrustes"):
            pck, auc = orc.keypoint_3d_pck_auc(xf, gf, None, al)
            assert abs(pck.item() - float(fx[f"{nm}.pck.{al}"])) < 1e-4 and abs(auc.item() - float(fx[f"{nm}.auc.{al}"])) < 1e-4
        pck, _ = orc.keypoint_3d_pck_auc(xf, gf, mask, "procrustes")
        assert abs(pck.item() - float(fx[f"{nm}.pck.procrustes.masked"])) < 1e-4
        np.testing.assert_allclose(orc.p_mpjpe(x, gt).item(), float(fx[f"{nm}.p_mpjpe"]), rtol=1e-6)
        pck, auc = orc.keypoint_3d_pck_auc(xf, gf, mask, "none")
        assert abs(pck.item() - float(fx[f"{nm}.pck.masked"])) < 1e-4 and abs(auc.item() - float(fx[f"{nm}.auc.masked"])) < 1e-4
        pck, _ = orc.keypoint_3d_pck_auc(xf, gf, None, "none", threshold=80.0)
        assert abs(pck.item() - float(fx[f"{nm}.pck.thr80"])) < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# input pipeline: oracle restatement of PoseSequenceGenerator + PoseFlip vs windows produced by the reference
# ---------------------------------------------------------------------------------------------------------------------
WINDOW_CASES = {"strided_drop": (False, True, False), "strided_pad": (False, False, False), "random_flip": (True, True, True),
                "strided_pad_flip": (False, False, True)}


def _window_sequences(fx):
    n = len(fx["lens"])
    return [fx[f"p3.{i}"] for i in range(n)], [fx[f"p2.{i}"] for i in range(n)]


@pytest.mark.parametrize("case", sorted(WINDOW_CASES))
def test_oracle_sequence_windows_match_reference_generator(case):
    fx = load_fixture("windows")
    p3, p2 = _window_sequences(fx)
    random_start, drop_last, flip = WINDOW_CASES[case]
    n = len(orc.window_tables([p.shape[0] for p in p3], 27, drop_last)[0])
    assert n == int(fx[f"{case}.len"])
    torch.manual_seed(2024)                                  # the same RNG stream the reference consumed item by item
    for i in range(n):
        x, y = orc.sequence_window(p3, p2, i, 27, random_start, drop_last, 0.5 if flip else None)
        np.testing.assert_array_equal(x.numpy(), fx[f"{case}.X"][i])
        np.testing.assert_array_equal(y.numpy(), fx[f"{case}.y"][i])


MISS_TYPES = ("random", "random_left_arm_right_leg", "structured_joint", "structured_frame", "noisy", "all")


@pytest.mark.parametrize("mt", MISS_TYPES)
def test_oracle_occlusion_patterns_match_reference_generator(mt):
    fx = load_fixture("windows")
    p3, p2 = _window_sequences(fx)
    torch.manual_seed(77)
    np.random.seed(99)
    for i in range(fx[f"miss.{mt}.X"].shape[0]):
        x, y = orc.sequence_window(p3, p2, i, 27, True, True, 0.5, miss_type=mt, miss_rate=0.3, noise_sigma=0.05)
        np.testing.assert_array_equal(x.numpy(), fx[f"miss.{mt}.X"][i])
        np.testing.assert_array_equal(y.numpy(), fx[f"miss.{mt}.y"][i])


# ---------------------------------------------------------------------------------------------------------------------------------
# dataset ingest (on-disk formats): oracle restatement vs what the reference's loaders produced from the same raw dictionaries
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,subjects,filt,stride", [("all", ["S1", "S9"], None, 1), ("walk_s2", ["S9", "S11"], ["walking"], 2),
                                                       ("s11_sit", ["S11"], ["sittingdown"], 1)])
def test_oracle_h36m_ingest_matches_reference_loaders(case, subjects, filt, stride):
    fx = np.load(GOLDEN + "/datasets.npz")
    h3, h2, _, _ = raw_dataset_dicts(fx)
    p3, p2, actions, cams = orc.h36m_sequences(h3, h2, h36m_calibration(), subjects, filt, stride)
    assert len(p3) == len(p2) == int(fx[f"h36m.{case}.n"]) and actions == list(fx[f"h36m.{case}.actions"])
    for i in range(len(p3)):
        np.testing.assert_array_equal(p2[i], fx[f"h36m.{case}.p2.{i}"])               # bit-exact: plain float32 / float64 arithmetic
        np.testing.assert_allclose(p3[i], fx[f"h36m.{case}.p3.{i}"], rtol=0, atol=2e-6)  # torch.cross may contract to FMA
        np.testing.assert_allclose(cams[i], fx[f"h36m.{case}.cam.{i}"], rtol=0, atol=0)
        assert np.all(p3[i][:, 0] == 0)


@pytest.mark.parametrize("split", ["train", "test"])
def test_oracle_3dhp_ingest_matches_reference_loader(split):
    fx = np.load(GOLDEN + "/datasets.npz")
    _, _, tr, te = raw_dataset_dicts(fx)
    p3, p2 = orc.hp3d_sequences(tr if split == "train" else te, split == "train")
    assert len(p3) == int(fx[f"hp.{split}.n"])
    for i in range(len(p3)):
        np.testing.assert_array_equal(p3[i], fx[f"hp.{split}.p3.{i}"])
        np.testing.assert_array_equal(p2[i], fx[f"hp.{split}.p2.{i}"])


def test_squared_loss_terms_and_grads():
    """train.sq_loss=True: squared WTA / velocity terms (losses.py:46-72,96-97,110-116) against the reference's own functions."""
    fx = load_fixture("loss_sq")
    cfg = dict(orc.DEFAULT_TRAIN_CFG, sq_loss=True)
    poses = torch.from_numpy(fx["poses"]).requires_grad_(True)
    scores = torch.from_numpy(fx["scores"]).requires_grad_(True)
    y = torch.from_numpy(fx["y"])
    total, terms = orc.rmcl_training_loss(poses, scores, y, cfg)
    total.backward()
    np.testing.assert_allclose([terms[k].item() for k in ("wloss", "score_reg", "vloss", "sreg")], fx["loss_terms"], rtol=1e-6)
    np.testing.assert_allclose(poses.grad.numpy(), fx["g_poses"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(scores.grad.numpy(), fx["g_scores"], rtol=1e-5, atol=1e-9)
    val, idx = orc.wta_l2_loss_and_activate_head(poses.detach(), y, torch.tensor(orc.STANDARD_H36M_WEIGHTS), squared=True)
    assert np.array_equal(idx.numpy(), fx["wta_idx"])
    np.testing.assert_allclose(val.numpy(), fx["wta_vals"], rtol=1e-6)
    for nm, w_loss in (("w", True), ("nw", False)):
        p1 = torch.from_numpy(fx["poses"][:, 0].copy()).requires_grad_(True)
        tot, t = orc.manifold_training_loss(p1, y, dict(cfg, w_loss=w_loss))
        tot.backward()
        np.testing.assert_allclose([t[k].item() for k in ("wloss", "vloss", "sreg")], fx[f"single_{nm}.terms"], rtol=1e-6)
        np.testing.assert_allclose(p1.grad.numpy(), fx[f"single_{nm}.g"], rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------------
# model.rot_dim=4: the 4-D rotation representation, decoder and whole models
# ---------------------------------------------------------------------------------------------------------------------------------
def test_rot4_decoder_matches_reference_and_known_answer():
    fx = load_fixture("decoder_rot4")
    rot = torch.from_numpy(fx["rot4d"]).requires_grad_(True)
    bl = torch.from_numpy(fx["bones"]).requires_grad_(True)
    poses = orc.pose_decoder(rot, bl)
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], rtol=1e-5, atol=1e-6)
    (poses * torch.from_numpy(fx["gpos"])).sum().backward()
    np.testing.assert_allclose(rot.grad.numpy(), fx["g_rot4d"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bl.grad.numpy(), fx["g_bones"], rtol=1e-4, atol=1e-6)
    ident = torch.tensor([0.0, 1.0, 1.0, 0.0]).repeat(1, 17, 1)            # (c1, s1, c2, s2) of the identity rotation
    np.testing.assert_allclose(orc.rotation_from_ortho4d(ident[0]).numpy(), np.broadcast_to(np.eye(3, dtype=np.float32), (17, 3, 3)), atol=0)
    tp = orc.pose_decoder(ident, torch.from_numpy(fx["tpose_lens"]))
    np.testing.assert_allclose(tp.numpy(), fx["tpose"], atol=1e-7)


def test_rot4_models_forward_loss_grads():
    fx = load_fixture("rmcl_tiny_rot4")
    st, poses, scores, total, terms = _run_rmcl(fx)
    assert st["rotations_module.head.0.prediction_head.weight"].shape[0] == 5
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], **TOL)
    np.testing.assert_allclose(scores.detach().numpy(), fx["scores"], **TOL)
    np.testing.assert_allclose(total.item(), float(fx["loss_total"]), rtol=1e-5)
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)
    fx = load_fixture("manifold_k1_rot4")
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    assert st["rotations_module.head.1.weight"].shape[0] == 4
    pred = orc.manifold_forward(torch.from_numpy(fx["X"]), st, orc.oracle_cfg(fx["cfg"]))
    np.testing.assert_allclose(pred.detach().numpy(), fx["poses"], **TOL)
    total, _ = orc.manifold_training_loss(pred, torch.from_numpy(fx["y"]))
    np.testing.assert_allclose(total.item(), float(fx["loss_total"]), rtol=1e-5)
    total.backward()
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_bare_mixste_forward_loss_grads():
    """model.arch=mixste (main_h36m_lifting.py:617-628): MixSTE.forward (mix_ste.py:175-191) + the single-hypothesis loss."""
    fx = load_fixture("mixste_tiny")
    T, C, depth, heads = [int(v) for v in fx["cfg_mixste"]]
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    pred = orc.mixste_forward(torch.from_numpy(fx["X"]), st, "", depth, heads)
    np.testing.assert_allclose(pred.detach().numpy(), fx["poses"], **TOL)
    total, terms = orc.manifold_training_loss(pred, torch.from_numpy(fx["y"]))
    np.testing.assert_allclose([terms[k].item() for k in ("wloss", "vloss", "sreg")], fx["loss_terms"], rtol=1e-5)
    total.backward()
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_rigid_segments_term_matches_reference():
    """train.rigid_seg_reg (make_loss :170-177): value, gradient wrt the prediction and parameter gradients of the 4-term total."""
    fx = load_fixture("mixste_tiny")
    T, C, depth, heads = [int(v) for v in fx["cfg_mixste"]]
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    pred = orc.mixste_forward(torch.from_numpy(fx["X"]), st, "", depth, heads)
    pred.retain_grad()
    r = 0.7 * orc.segments_time_consistency(pred.permute(0, 3, 2, 1), "sum")
    r.backward(retain_graph=True)
    np.testing.assert_allclose(r.item(), float(fx["rigid_term"]), rtol=1e-5)
    np.testing.assert_allclose(pred.grad.numpy(), fx["rigid_g_pred"], rtol=1e-4, atol=1e-8)
    for v in st.values():
        v.grad = None
    pred = orc.mixste_forward(torch.from_numpy(fx["X"]), st, "", depth, heads)
    total, terms = orc.manifold_training_loss(pred, torch.from_numpy(fx["y"]), dict(orc.DEFAULT_TRAIN_CFG, rigid_seg_reg=0.7))
    np.testing.assert_allclose(total.item(), float(fx["rigid_total"]), rtol=1e-5)
    total.backward()
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g_rigid::" + k], rtol=2e-4, atol=2e-6, err_msg=k)

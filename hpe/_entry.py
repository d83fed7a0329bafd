"""Shared implementation of the two lifting entry points (counterparts of the reference's Hydra scripts
hpe/main_h36m_lifting.py:711-1270 and hpe/main_3dhp.py:662-1063): same `group.key=value` override grammar, same config
keys (hpe/conf/config.yaml), same model construction (_instantiate_model :613-670), loss assembly (make_loss :101-178),
optimizer / scheduler (:227-270), checkpoint names (save_state :75-98: model{tag}.pth / params{tag}.pth) and the MPJPE
evaluation of evaluate() (hpe/eval_utils.py:16-223: weighted-average, best-score and oracle aggregation, millimetres).
With `data.data_dir` the reference's files (data_3d_h36m.npz + data_2d_h36m_<keypoints>.npz, or data_{train,test}_3dhp.npz) are
ingested on the device (manipose_amd/data/ingest.py) and stay resident in HBM; without it the script trains / evaluates on synthetic
H36M-shaped sequences so that the whole MI355X path (window kernel, engine, fused loss, RCCL data parallelism, fused Adam) runs.
Launch on N GPUs with `python -m torch.distributed.run --nproc-per-node N hpe/main_h36m_lifting.py ...`.
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


from manipose_amd.hydra_lite import Cfg, parse_value  # noqa: E402,F401
from manipose_amd import hydra_lite  # noqa: E402


def load_config(argv, extra_defaults=None):
    """hpe/conf/config.yaml + overrides in the reference's Hydra grammar (main_h36m_lifting.py:711, README.md:54-111): see
    manipose_amd/hydra_lite.py.  Both evaluation commands of the reference's README parse unchanged."""
    return hydra_lite.load_config(os.path.join(ROOT, "hpe", "conf"), argv, extra_defaults)


def instantiate_model(cfg):
    from manipose_amd import ManifoldMixSTE, MixSTE, RMCLManifoldMixSTE, h36m_skeleton
    sk = h36m_skeleton()
    if cfg.model.arch == "mixste":                          # main_h36m_lifting.py:617-628
        model = MixSTE(num_frame=cfg.data.seq_len, num_joints=sk.num_joints, in_chans=2, out_dim=3, num_heads=cfg.model.nheads,
                       depth=cfg.model.layers, embed_dim=cfg.model.channels, drop_path_rate=cfg.model.drop_path_rate, mup=cfg.model.mup)
        model.precision = cfg.model.precision
        _engine_options(model, cfg)
        return model
    kw = dict(skeleton=sk, num_frame=cfg.data.seq_len, num_joints=sk.num_joints, num_bones=sk.num_bones, in_chans=2,
              rot_rep_dim=cfg.model.rot_dim, num_heads_rot=cfg.model.nheads, depth_rot=cfg.model.layers,
              embed_dim_rot=cfg.model.channels, num_heads_seg=cfg.model.nheads_seg, depth_seg=cfg.model.layers_seg,
              embed_dim_seg=cfg.model.channels_seg, drop_path_rate=cfg.model.drop_path_rate, mup=cfg.model.mup)
    if cfg.model.arch == "rmcl_manifold":
        model = RMCLManifoldMixSTE(n_hyp=cfg.multi_hyp.n_hyp, **kw)
    elif cfg.model.arch == "manifold":
        model = ManifoldMixSTE(**kw)
    else:
        raise ValueError(f"Only MixSTE, Manifold-MixSTE and RMCL-Manifold-MixSTE implemented for now. Got option {cfg.model.arch}.")
    model.precision = cfg.model.precision
    _engine_options(model, cfg)
    return model


def _engine_options(model, cfg):
    """model.f16f8 / model.f16_backward of the config (mp_model_config, ABI v7): the operand form of the qkv / fc1 (/ fc2) layers of a
    bf16x3 model (ignored for the other precisions).  Config default since round 6: f16f8 = 3 (all four Linear layers of a block as one fp16 + one
    block-scaled fp8 product; a model that does not qualify runs three bf16 products everywhere), bf16 backward."""
    bf16x3 = str(getattr(cfg.model, "precision", "")) == "bf16x3"
    model.f16f8 = int(getattr(cfg.model, "f16f8", 0)) if bf16x3 else 0
    model.f16_backward = bool(getattr(cfg.model, "f16_backward", False)) if bf16x3 else False


def synthetic_windows(n, T, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    X = (0.3 * torch.randn(n, T, 17, 2, device=device, generator=g)).clamp(-1, 1)
    y = 0.3 * torch.randn(n, T, 17, 3, device=device, generator=g)
    y[:, :, 0] = 0
    return X, y


def synthetic_generator(cfg, device, seed, rank, world):
    """The training data path of the reference (fetch -> PoseSequenceGenerator(random_start=True, PoseFlip(0.5)) -> DataLoader,
    main_h36m_lifting.py:560-610) with the sequences resident in HBM: synthetic H36M-shaped action sequences (no dataset ships here),
    every rank keeps its own shard of them."""
    from manipose_amd import h36m_skeleton
    from manipose_amd.augmentations import PoseFlip
    from manipose_amd.data import PoseSequenceGenerator
    from manipose_amd.distributed import shard_windows
    rng = np.random.default_rng(seed)
    n_seq = max(world, int(cfg.data.get("synthetic_sequences", 16)))
    lens = [int(cfg.data.seq_len) * 4 + 37 * (i % 5) + 11 for i in range(n_seq)]
    mine = list(shard_windows(n_seq, rank, world))
    p3, p2 = [], []
    for i in range(n_seq):
        a = (0.3 * rng.standard_normal((lens[i], 17, 3))).astype(np.float32)
        b = np.clip(0.3 * rng.standard_normal((lens[i], 17, 2)), -1, 1).astype(np.float32)
        a[:, 0] = 0
        if i in mine:
            p3.append(a); p2.append(b)
    tf = PoseFlip(h36m_skeleton(), 0.5) if cfg.train.flip_aug else None
    return PoseSequenceGenerator(p3, p2, None, seq_len=int(cfg.data.seq_len), random_start=True, drop_last=True,
                                 miss_type=cfg.data.get("miss_type", "no_miss"), miss_rate=cfg.data.get("miss_rate", 0.2),
                                 transform=tf, device=device)


def load_sequences(cfg, device):
    """fetch_and_prepare_data + get_subjects_and_actions + the fetch() calls of create_dataloader (main_h36m_lifting.py:511-583,
    main_3dhp.py:503-532): {"train" | "valid": (poses_3d, poses_2d, cameras), "test": {group: (...)}} as lists of device tensors.
    The reference's pickle cache of the prepared dataset is not needed: the ingest is a few hundred kernel launches."""
    from manipose_amd.data import Dataset3DHP, Human36mDataset, create_2d_data, fetch, read_3d_data
    from manipose_amd.data.ingest import TEST_SUBJECTS, TRAIN_SUBJECTS
    root = str(cfg.data.data_dir)
    with torch.cuda.device(device):
        if cfg.data.dataset == "3dhp":
            out = {"test": {}}
            if cfg.run.train:
                tr = Dataset3DHP(cfg, root + "/", train=True, device=device)
                out["train"] = (tr.poses, tr.poses_2d, None)
            te = Dataset3DHP(cfg, root + "/", train=False, device=device)
            out["valid"] = (te.poses, te.poses_2d, None)
            out["test"]["all"] = out["valid"]
            return out
        ds = read_3d_data(Human36mDataset(os.path.join(root, f"data_3d_{cfg.data.dataset}.npz"), n_joints=cfg.data.joints, device=device))
        kp = create_2d_data(os.path.join(root, f"data_2d_{cfg.data.dataset}_{cfg.data.keypoints}.npz"), ds)
        if cfg.data.use_valid:
            s_train, s_val = TRAIN_SUBJECTS[:-1], TRAIN_SUBJECTS[-1:]
        else:
            s_train, s_val = TRAIN_SUBJECTS, TEST_SUBJECTS
        if cfg.data.data == "one":
            s_train = [s_train[0]]
        actions = None if cfg.data.actions == "*" else [ds.define_actions(a)[0] for a in str(cfg.data.actions).split(",")]
        have = lambda subjects: [s for s in subjects if s in kp]
        out = {"train": fetch(have(s_train), ds, kp, actions)[:2] + (None,), "valid": fetch(have(s_val), ds, kp, actions)[:2] + (None,), "test": {}}
        for a in (actions or ds.define_actions()):                 # per-action test on S11, main_h36m_lifting.py:884-893
            p3, p2, _, _ = fetch(have(["S11"]), ds, kp, [a])
            if p3:
                out["test"][a] = (p3, p2, None)
        return out


def window_generator(cfg, seqs, train, device):
    """create_dataloader (main_h36m_lifting.py:569-610) minus the DataLoader: the generator itself serves whole batches."""
    from manipose_amd import h36m_skeleton
    from manipose_amd.augmentations import PoseFlip
    from manipose_amd.data import PoseSequenceGenerator
    tf = PoseFlip(h36m_skeleton(), 0.5) if cfg.train.flip_aug else None
    return PoseSequenceGenerator(seqs[0], seqs[1], seqs[2], seq_len=int(cfg.data.seq_len), random_start=train, miss_type=cfg.data.miss_type,
                                 miss_rate=cfg.data.miss_rate, noise_sigma=cfg.data.get("noise_sigma", 5), transform=tf, device=device)


def epoch_batches(gen, batch, shuffle, rank=0, world=1, seed=None, pad=False):
    """DataLoader(shuffle, drop_last=False) over the generator's indices, dealt round-robin over the ranks (DistributedSampler-style:
    every rank draws the same permutation from ``seed`` and keeps every world-th index).  ``pad=True`` (training) wraps the order
    around to a multiple of ``world`` so that every rank runs the same number of steps - each step holds a collective."""
    n = len(gen)
    if shuffle:
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    if pad and world > 1 and n % world:
        order = order + order[:world - n % world]
    order = order[rank::world]
    for i in range(0, len(order), batch):
        yield gen.batch(order[i:i + batch])


def tensor_batches(X, y, batch):
    for i in range(0, X.shape[0], batch):
        yield X[i:i + batch], y[i:i + batch]


@torch.no_grad()
def evaluate(model, X, y=None, batch=None, tta=True, analytics=False, distributed=False):
    """MPJPE (mm) of the aggregated / best-score / oracle hypotheses, with the reference's flip test-time augmentation
    (hpe/eval_utils.py:16-223).  The flipped copy is batched with the original into ONE forward of 2B windows
    instead of a second pass.  ``analytics=True`` adds the reference's evaluation table (main_h36m_lifting.py:933-990,
    main_3dhp.py:860-910: MPSSE, MPSCE, segment-length error, MSE / error variance, 3DPCK, AUC, per-joint errors) of the
    aggregated prediction in millimetres, from the one-pass HIP analytics kernel.  ``distributed=True``: every rank
    evaluates its own share of the batches and the error sums / frame counts are sum-reduced at the end."""
    from manipose_amd import RMCLManifoldMixSTE
    from manipose_amd.augmentations import pose_flip
    from manipose_amd.metrics import mpjpe_error
    from manipose_amd.metrics.analytics import AnalyticsAccumulator, pose_analytics, procrustes_sums
    acc = AnalyticsAccumulator() if analytics else None
    model.eval()
    sk = model.decoder.skeleton if hasattr(model, "decoder") else _h36m()
    rmcl = isinstance(model, RMCLManifoldMixSTE)
    sums = {"mpjpe": 0.0, "ps_oracle_mpjpe": 0.0, "oracle_mpjpe": 0.0}
    n = 0
    for xb, yb in (tensor_batches(X, y, batch) if torch.is_tensor(X) else X):      # tensors, or any iterable of (X, y) batches
        nb = xb.shape[0]
        if tta:
            xin = torch.cat([xb, pose_flip((xb.clone(),), sk)[0]], dim=0)
        else:
            xin = xb
        out = model(xin)
        if rmcl:
            poses, scores = out
            pred = model.aggregate(poses[:nb], scores[:nb], "weighted_ave")
            best = model.aggregate(poses[:nb], scores[:nb], "best_score")
            orac = model.aggregate(poses[:nb], mode="oracle", ground_truth=yb)[1]
            if tta:
                hyp_f = pose_flip((poses[nb:].clone(),), sk)[0]          # flipped hypotheses mapped back to the original frame
                pred = (pred + model.aggregate(hyp_f, scores[nb:], "weighted_ave")) / 2
                best = (best + model.aggregate(hyp_f, scores[nb:], "best_score")) / 2
                orac = (orac + model.aggregate(hyp_f, mode="oracle", ground_truth=yb)[1]) / 2
            sums["mpjpe"] += mpjpe_error(pred, yb, "sum").item()
            sums["ps_oracle_mpjpe"] += mpjpe_error(best, yb, "sum").item()
            sums["oracle_mpjpe"] += mpjpe_error(orac, yb, "sum").item()
        else:
            pred = out[:nb]
            if tta:
                pred = (pred + pose_flip((out[nb:].clone(),), sk)[0]) / 2
            sums["mpjpe"] += mpjpe_error(pred, yb, "sum").item()
        n += yb.numel() // 3
        if acc is not None:
            acc.add(pose_analytics(pred.detach().contiguous(), yb.contiguous(), pred_scale=1000.0, gt_scale=1000.0))
            acc.add_procrustes(procrustes_sums(pred.detach(), yb, pred_scale=1000.0, gt_scale=1000.0))
    if distributed:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([sums["mpjpe"], sums["ps_oracle_mpjpe"], sums["oracle_mpjpe"], float(n)], dtype=torch.float64, device=sk_device(model))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            sums = {"mpjpe": t[0].item(), "ps_oracle_mpjpe": t[1].item(), "oracle_mpjpe": t[2].item()}
            n = t[3].item()
            if acc is not None:
                acc.all_reduce()
    out = {k: 1000.0 * v / n for k, v in sums.items() if v > 0}
    if acc is not None:
        out["analytics"] = acc.report()
    return out


def _h36m():
    from manipose_amd import h36m_skeleton
    return h36m_skeleton()


def sk_device(model):
    return next(model.parameters()).device


def save_state(model, trainer, scheduler_state, epoch, folder, tag=None):
    tag = f"_{tag}" if tag else ""
    torch.save(model.state_dict(), os.path.join(folder, f"model{tag}.pth"))
    torch.save({"optimizer": trainer.opt.state_dict(), "scheduler": scheduler_state, "epoch": epoch},
               os.path.join(folder, f"params{tag}.pth"))


def run(argv, extra_defaults=None):
    from manipose_amd.distributed import broadcast_parameters, init_from_env
    from manipose_amd.training import LiftingTrainer
    cfg = load_config(argv, extra_defaults)
    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("the lifting entry points need an MI355X (ROCm device); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(cfg.run.seed)
    model = instantiate_model(cfg)
    mup_mults = None
    if cfg.model.mup:                          # create_model / set_mup_base_shapes (main_h36m_lifting.py:673-708): BEFORE any checkpoint is loaded,
        # as in the reference - set_base_shapes rescales the freshly initialised MuReadout weights by sqrt(width_mult), which must never
        # touch weights that come from a checkpoint
        from manipose_amd.mup_lite import make_base_shapes, mu_init_params, mup_lr_multipliers, set_base_shapes
        small, big = copy.deepcopy(cfg), copy.deepcopy(cfg)
        small["model"]["channels"], small["model"]["channels_seg"], small["data"]["seq_len"] = 64, 64, 27
        big["model"]["channels"], big["model"]["channels_seg"], big["data"]["seq_len"] = 128, 128, 81
        set_base_shapes(model, make_base_shapes(instantiate_model(Cfg.wrap(small)), instantiate_model(Cfg.wrap(big))))
        mup_mults = mup_lr_multipliers(model)
    if cfg.run.checkpoint_model:               # main_h36m_lifting.py:754-760: checkpoint weights are used untouched
        ck = torch.load(cfg.run.checkpoint_model, map_location="cpu")
        model.load_state_dict(ck["model_pos"] if "model_pos" in ck else ck)
    elif cfg.model.mup:                        # :761-764: re-initialisation according to muP only without a checkpoint
        mu_init_params(model)
    model.max_batch_hint = max(cfg.train.batch_size, 2 * cfg.train.batch_size_test)      # x2: flip-TTA batches the mirrored copy
    model = model.to(dev)
    if cfg.train.get("lat_sym_regularization", 0) > 0:
        print("warning: Lateral symmetry regularization is not implemented yet!", flush=True)      # as main_h36m_lifting.py:109-110
    trainer = LiftingTrainer(model, lr=cfg.train.lr, weight_decay=1e-6, w_loss=cfg.train.w_loss, vel_loss=cfg.train.vel_loss,
                             smooth_reg=cfg.train.smooth_reg, rmcl_score_reg=cfg.train.rmcl_score_reg, seed=cfg.run.seed,
                             sq_loss=cfg.train.sq_loss, rigid_seg_reg=cfg.train.get("rigid_seg_reg", 0.0))
    if mup_mults is not None:                  # MuAdam (main_h36m_lifting.py:227-232): matrix-like parameters train with lr / width_mult
        trainer.opt.set_multipliers(mup_mults)
    broadcast_parameters(model.flat_parameters())
    start_epoch, sched_state = 0, None
    if cfg.run.checkpoint_params:
        st = torch.load(cfg.run.checkpoint_params, map_location="cpu")
        trainer.opt.load_state_dict(st["optimizer"])
        trainer.step_no = trainer.opt.step_count          # DropPath masks are a function of (seed, step): continue, do not replay
        start_epoch = st["epoch"]
        sched_state = st.get("scheduler")
    T, B, Bt = cfg.data.seq_len, cfg.train.batch_size, cfg.train.batch_size_test
    out_dir = os.path.join(os.getcwd(), cfg.run.experiment)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    real = bool(cfg.data.data_dir)
    # evaluation is sharded over the ranks like training whenever every rank gets at least one window; the error sums and
    # frame counts are sum-reduced inside evaluate(); otherwise rank 0 evaluates alone
    def sharded(gen):
        share = world > 1 and len(gen) >= world
        return (lambda: epoch_batches(gen, Bt, shuffle=False, rank=rank if share else 0, world=world if share else 1)), share

    def allsum(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t)
        return t.item()

    if real:
        seqs = load_sequences(cfg, dev)
        gen_valid = window_generator(cfg, seqs["valid"], False, dev)
        valid_batches, valid_shared = sharded(gen_valid)
        if rank == 0:
            print(f">>> Validation dataset length: {len(gen_valid)} windows of {T} frames", flush=True)
    else:
        Xv, yv = synthetic_windows(4 * Bt, T, dev, seed=10_000)
        valid_shared = world > 1 and Xv.shape[0] >= world
        valid_batches = (lambda: tensor_batches(Xv[rank::world], yv[rank::world], Bt)) if valid_shared else (lambda: tensor_batches(Xv, yv, Bt))
    from manipose_amd.optim import make_lr_scheduler
    sched = make_lr_scheduler(trainer.opt, cfg.train.lr_scheduler, cfg.train.epochs, cfg.train.n_annealing, cfg.train.lr_min,
                              cfg.train.lr_patience, cfg.train.lr_threshold)            # main_h36m_lifting.py:243-264
    if sched_state is not None and sched_state.get("kind") == sched.state_dict()["kind"]:
        sched.load_state_dict(sched_state)
    best_val, best_eval = 1e10, {}
    best_model_state = None                  # main_h36m_lifting.py:390,491: weights of the best validation loss / best MPJPE, reloaded before the test
    train_curve, valid_curve = [], []
    if cfg.run.train:
        if real:
            gen = window_generator(cfg, seqs["train"], True, dev)       # sequences resident in HBM, one gather kernel per batch
            if rank == 0:
                print(f">>> Training dataset length: {len(gen)} windows of {T} frames", flush=True)
        else:
            gen = synthetic_generator(cfg, dev, cfg.run.seed, rank, world)
        torch.manual_seed(cfg.run.seed + 1000 * rank)                        # per-rank window / flip draws
        np.random.seed(cfg.run.seed + 1000 * rank)                           # per-rank occlusion draws
        for epoch in range(start_epoch, cfg.train.epochs):
            model.train()
            acc = torch.zeros(4, device=dev)
            if real:       # one pass over the windows, shuffled, dealt over the ranks (main_h36m_lifting.py:597-610)
                batches = epoch_batches(gen, B, shuffle=True, rank=rank, world=world, seed=cfg.run.seed + epoch, pad=True)
            else:          # shuffled sampling with replacement over this rank's synthetic windows
                batches = (gen.batch(torch.randint(0, len(gen), (B,)).tolist()) for _ in range(cfg.train.steps_per_epoch))
            steps = 0
            for X, y in batches:
                acc += trainer.train_step(X, y)                      # device-side accumulation: no per-step host sync
                steps += 1
            terms = (acc / max(steps, 1)).tolist()
            train_curve.append(sum(terms))
            if (epoch + 1) % cfg.train.valid_epoch_interval == 0:
                model.eval()
                val = sum(trainer.eval_loss(xb, yb).sum().item() for xb, yb in valid_batches())
                if valid_shared:
                    val = allsum(val)
                valid_curve.append(val)
                if best_val > val:                                     # main_h36m_lifting.py:374-398
                    best_val = val
                    best_model_state = copy.deepcopy(model.state_dict())
                    if rank == 0:
                        save_state(model, trainer, sched.state_dict(), epoch, out_dir, "best_val")
                if cfg.train.lr_scheduler == "plateau":                # :400-403: the plateau scheduler watches the running best
                    sched.step(best_val)
                else:
                    sched.step()
            if rank == 0:
                names = ("wloss", "score_reg", "vloss", "sreg") if trainer.rmcl else ("wloss", "vloss", "sreg", "rigid_seg_reg")
                print(f"epoch {epoch}: tr_loss {sum(terms):.5f} " + " ".join(f"{n} {v:.5f}" for n, v in zip(names, terms))
                      + f" | best val {best_val:.5f} lr {sched.get_last_lr()[0]:.2e}", flush=True)
            if (epoch + 1) % cfg.train.mpjpe_epoch_interval == 0 and (rank == 0 or valid_shared):      # main_h36m_lifting.py:405-470
                ev = evaluate(model, valid_batches(), tta=cfg.train.tta, distributed=valid_shared)
                if rank == 0:
                    print("   eval:", {k: round(v, 3) for k, v in ev.items()}, flush=True)
                for key, tag in (("mpjpe", "best_mpjpe"), ("oracle_mpjpe", "best_oracle_mpjpe"), ("ps_oracle_mpjpe", "best_ps_oracle_mpjpe")):
                    if key in ev and ev[key] < best_eval.get(key, 1e10):           # tags of main_h36m_lifting.py:440-489
                        best_eval[key] = ev[key]
                        if key == "mpjpe":
                            best_model_state = copy.deepcopy(model.state_dict())
                        if rank == 0:
                            save_state(model, trainer, sched.state_dict(), epoch, out_dir, tag)
        if rank == 0:
            save_state(model, trainer, sched.state_dict(), cfg.train.epochs, out_dir, "end")
            np.save(os.path.join(out_dir, "train_loss.npy"), np.array(train_curve))           # :503-504
            np.save(os.path.join(out_dir, "valid_loss.npy"), np.array(valid_curve))
        if best_model_state is not None:                   # "load best weights" (main_h36m_lifting.py:506-507): the test below runs on them
            model.load_state_dict(best_model_state)
    if cfg.run.test:
        groups = {"synthetic": (valid_batches, valid_shared)}
        if real:        # per action (H36M: subject S11, main_h36m_lifting.py:884-990) or the whole 3DHP test set (main_3dhp.py:800-910)
            groups = {name: sharded(window_generator(cfg, sq, False, dev)) for name, sq in seqs["test"].items()}
        rows = {}
        for name, (make, shared) in groups.items():
            if rank != 0 and not shared:
                continue
            res = evaluate(model, make(), tta=cfg.train.tta, analytics=True, distributed=shared)
            table = res.pop("analytics")
            rows[name] = res
            if rank == 0:
                print(f"test [{name}]:", {k: round(v, 3) for k, v in res.items()}, flush=True)
                print(f"test analytics [{name}] (mm):", {k: round(v, 4) for k, v in table.items() if not isinstance(v, list)}, flush=True)
        if len(rows) > 1 and rank == 0:
            keys = sorted({k for r in rows.values() for k in r})
            print("test [average over groups]:", {k: round(float(np.mean([r[k] for r in rows.values() if k in r])), 3) for k in keys}, flush=True)
    return best_val

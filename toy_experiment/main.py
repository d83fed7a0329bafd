"""Toy experiment of the ManiPose paper: lifting a 1-D abscissa to the 2-D point on a circle with (a) a plain MLP, (b) the
manifold-constrained MLP, (c) its multi-hypothesis rMCL version.  Counterpart of the reference's toy_experiment/main.py:25-327 for the
1-D -> 2-D scenarios (BASELINE config #1; everything runs on the CPU, no GPU kernel is involved):
    cd toy_experiment ; python main.py model.arch=constrained +train=constrained_easy
Same override grammar and config keys as the reference (conf/config.yaml, conf/train/*.yaml).  Writes <cwd>/outputs/<run.experiment>/
{model_best_val.pth, params_best_val.pth, train_loss.npy, test_predictions.npy}; returns the validation MPJPE."""
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from circle_toy import (ConstrainedMlp, ConstrainedMlpRmcl, LiftingDataset, Mlp, SquaredReLU, Trainer, calc_mpjpe, distance_to_circle,  # noqa: E402
                        oracle_multihyp_mpjpe, scenario)
from manipose_amd.hydra_lite import load_config  # noqa: E402

ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sqrelu": SquaredReLU}


def build_model(cfg):
    if cfg.diffusion.enabled:
        raise NotImplementedError("the diffusion baseline of the reference's toy experiment is not part of this repository")
    if cfg.model.act not in ACTIVATIONS:
        raise ValueError(f"Currently supported activations are 'relu' and 'tanh'.Got {cfg.model.act}.")
    act = ACTIVATIONS[cfg.model.act]
    kw = dict(hidden_features=cfg.model.hidden_features, n_layers=cfg.model.layers, act_layer=act)
    if cfg.model.arch == "mlp":
        return Mlp(in_features=1, out_features=2, **kw)
    if cfg.model.arch == "constrained":
        return ConstrainedMlp(in_features=1, out_features=1, radius=cfg.data.radius, **kw)
    if cfg.model.arch == "constrained_rmcl":
        return ConstrainedMlpRmcl(in_features=1, out_features=1, radius=cfg.data.radius, n_hyp=cfg.multi_hyp.nsamples, beta=cfg.model.beta, **kw)
    raise ValueError(f"Possible 'arch' values are 'mlp' and 'constrained'.Got {cfg.model.arch}.")


def main(argv=None):
    cfg = load_config(os.path.join(HERE, "conf"), sys.argv[1:] if argv is None else argv)
    if "3D" in str(cfg.data.scenario):
        raise SystemExit("data.scenario=torus-2Dto3D samples from pyro's SineBivariateVonMises, which is not available here; "
                         "the 1-D -> 2-D scenarios are: easy, hard-1, hard-2, hard-4")
    out_dir = os.path.join(os.getcwd(), "outputs", str(cfg.run.experiment))
    os.makedirs(out_dir, exist_ok=True)
    random.seed(cfg.run.seed)
    np.random.seed(cfg.run.seed)
    torch.manual_seed(cfg.run.seed)
    dist = scenario(cfg.data.scenario, cfg.data.radius, cfg.run.seed)
    data = LiftingDataset(dist, cfg.data.n_train, cfg.data.n_val, cfg.data.n_test)
    loader = data.get_tr_loader(batch_size=cfg.train.batch_size, num_workers=cfg.train.workers)
    model = build_model(cfg)
    if cfg.train.optim not in ("adam", "sgd"):
        raise ValueError(f"Currently supported optim_cls values are 'adam' and 'sgd'.Got {cfg.train.optim}.")
    trainer = Trainer(model=model, optim_cls=torch.optim.Adam if cfg.train.optim == "adam" else torch.optim.SGD,
                      sched_cls=torch.optim.lr_scheduler.ReduceLROnPlateau if cfg.train.lr_scheduler else None, checkpointing_dir=out_dir,
                      lr=cfg.train.lr, config_train=cfg.train, device="cpu", config_data=cfg.data)
    if cfg.run.train:
        trainer.train(epochs=cfg.train.epochs, loader=loader, loss_func=F.mse_loss, val_data=data.validation_set)
        np.save(os.path.join(out_dir, "train_loss.npy"), np.array(trainer.loss_list))
    val_mpjpe = None
    if cfg.run.test:
        sets = (data.validation_set, data.test_set)
        (val_mpjpe, test_mpjpe), (_, test_pred), hyps = trainer.eval(sets, calc_mpjpe)
        (val_dtc, test_dtc), _, _ = trainer.eval(sets, distance_to_circle)
        res = {"val_mpjpe": val_mpjpe, "test_mpjpe": test_mpjpe, "val_dtc": val_dtc, "test_dtc": test_dtc}
        if hyps is not None:
            res["test_oracle_mpjpe"] = oracle_multihyp_mpjpe(hyps[1], data.Y_test)
        print(" ".join(f"{k} : {v:.5f}" for k, v in res.items()), flush=True)
        np.save(os.path.join(out_dir, "test_predictions.npy"), test_pred.numpy())
    return val_mpjpe


if __name__ == "__main__":
    main()

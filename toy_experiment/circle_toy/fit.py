"""Training loop and metrics of the toy experiment.  Counterpart of the reference's toy_experiment/training/{trainer.py:20-234,299-308,
metrics.py:5-29, averager.py}.  Faithful to one property of the reference that matters for parity: the trainer never switches the model
to eval mode, so BatchNorm normalises every batch - validation and test sets included - with that batch's own statistics."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from .networks import ConstrainedMlpRmcl


def calc_mpjpe(pred: torch.Tensor, gt: torch.Tensor) -> float:
    return (pred - gt).norm(dim=1).mean().item()


def distance_to_circle(pred: torch.Tensor) -> float:
    """1 - mean ||pred||: how far inside the unit circle the predictions fall (0 on the manifold)."""
    return 1 - pred.norm(dim=1).mean().item()


def oracle_multihyp_mpjpe(hypothesis: torch.Tensor, gt: torch.Tensor) -> float:
    return (hypothesis[..., :2] - gt[:, None, :]).norm(dim=2).min(dim=1).values.mean().item()


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class Trainer:
    def __init__(self, model, checkpointing_dir, config_train, optim_cls=torch.optim.Adam, sched_cls=None, lr: float = 1e-3, device="cpu",
                 config_data=None):
        self.model = model.to(device)
        self.device = device
        self.lr = lr
        self.checkpointing_dir = Path(checkpointing_dir)
        self.mcl_enabled = isinstance(model, ConstrainedMlpRmcl)
        self.optim = optim_cls(model.parameters(), lr=lr)
        self.scheduler = None if sched_cls is None else sched_cls(optimizer=self.optim, mode="min", factor=0.5, min_lr=config_train.lr_min,
                                                                  patience=config_train.lr_patience, threshold=config_train.lr_threshold)
        self.reset_metrics()

    def reset_metrics(self):
        self.loss_list, self.val_loss_list, self.loss_accum = [], [], AverageMeter()

    def _loss(self, x, y, loss_func):
        out = self.model(x)
        return self.model.wta_with_scoring_l2_loss(out, y) if self.mcl_enabled else loss_func(out, y)      # rMCL models bring their own loss

    def train(self, epochs: int, loader: DataLoader, loss_func, val_data: TensorDataset = None, log=print):
        best = np.inf
        for epoch in range(1, epochs + 1):
            self.loss_accum.reset()
            for x, y in loader:
                x, y = x.to(self.device), y.to(self.device)
                self.optim.zero_grad()
                loss = self._loss(x, y, loss_func)
                loss.backward()
                self.optim.step()
                self.loss_accum.update(loss.item(), n=loader.batch_size)
            self.loss_list.append(self.loss_accum.avg)
            if val_data is not None:
                val = self.eval((val_data,), loss_func)[0][0].item()      # rMCL models: loss_func of the score-weighted aggregate, as the reference
                self.val_loss_list.append(val)
                if val < best:
                    best = val
                    self.save_state(epoch, "best_val")
                if self.scheduler is not None:
                    self.scheduler.step(best)                                           # the plateau scheduler watches the running best
            if log is not None:
                log(f"epoch {epoch}: loss {self.loss_list[-1]:.5f}" + (f" val {self.val_loss_list[-1]:.5f}" if val_data is not None else ""))
        ck = self.checkpointing_dir / "model_best_val.pth"
        if ck.exists():                                                                  # test on the weights of the best validation loss
            self.model.load_state_dict(torch.load(ck))

    def eval(self, eval_sets, metric, raw_hypotheses: bool = False):
        """-> (performances, predictions, hypotheses or None), one entry per set.  Multi-hypothesis models are scored on their aggregated
        (score-weighted) prediction unless raw_hypotheses is set; a metric of one argument is called on the predictions alone."""
        perfs, preds, hyps = [], [], ([] if self.mcl_enabled else None)
        with torch.no_grad():
            for ds in eval_sets:
                X, y = (t.to(self.device) for t in ds.tensors)
                out = self.model(X)
                if self.mcl_enabled:
                    hyps.append(out)
                    if not raw_hypotheses:
                        out = self.model.aggregate(out)
                preds.append(out)
                try:
                    perfs.append(metric(out, y))
                except TypeError:
                    perfs.append(metric(out))
        return perfs, preds, hyps

    def save_state(self, epoch_no: int, tag: str = None):
        tag = f"_{tag}" if tag else ""
        params = {"optimizer": self.optim.state_dict(), "epoch": epoch_no}
        if self.scheduler is not None:
            params["scheduler"] = self.scheduler.state_dict()
        torch.save(self.model.state_dict(), self.checkpointing_dir / f"model{tag}.pth")
        torch.save(params, self.checkpointing_dir / f"params{tag}.pth")

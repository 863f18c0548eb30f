"""Host-side training/eval harness surface of the reference (mst/models/base_model.py:10-181).

Same class names, constructor arguments, step/optimizer/checkpoint helpers.  pytorch_lightning and
torchmetrics are optional: when present the classes derive from ``pl.LightningModule`` and use the
real metrics (so ``Trainer.fit`` / ``load_from_checkpoint`` work unchanged); when absent (this
image) a minimal stand-in supplies ``.device``, ``save_hyperparameters``, ``log`` and
``load_from_checkpoint`` so inference scripts and tests run as plain ``nn.Module``s.
"""
from __future__ import annotations

import inspect
import json
from pathlib import Path

import torch
import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _LightningBase = pl.LightningModule
    HAVE_LIGHTNING = True
except ImportError:
    HAVE_LIGHTNING = False

    class _LightningBase(nn.Module):
        """The few LightningModule facilities the MST classes rely on."""

        def __init__(self):
            super().__init__()
            self.hparams = {}
            self.logged = {}

        def save_hyperparameters(self, *args, **kwargs):
            frame = inspect.currentframe().f_back
            # climb to the outermost __init__ of this object (the concrete class's)
            init_locals = None
            while frame is not None:
                if frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self:
                    init_locals = frame.f_locals
                frame = frame.f_back
            if init_locals is not None:
                hp = {k: v for k, v in init_locals.items() if k not in ("self", "__class__") and not k.startswith("_")}
                kw = hp.pop("kwargs", {})
                hp.update(kw if isinstance(kw, dict) else {})
                self.hparams = hp

        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model

try:  # pragma: no cover
    from torchmetrics import AUROC, Accuracy
except ImportError:
    class _Metric(nn.Module):
        def __init__(self, task="multiclass", num_classes=2, **kwargs):
            super().__init__()
            self.num_classes = num_classes
            self.reset()

        def reset(self):
            self._pred, self._target = [], []

        def update(self, pred, target):
            self._pred.append(pred.detach().float().cpu())
            self._target.append(target.detach().cpu())

    class Accuracy(_Metric):
        def compute(self):
            if not self._pred:
                return torch.tensor(float("nan"))
            p, t = torch.cat(self._pred), torch.cat(self._target)
            return (p.argmax(-1) == t).float().mean()

    class AUROC(_Metric):
        """Macro one-vs-rest AUROC by the rank statistic."""

        def compute(self):
            if not self._pred:
                return torch.tensor(float("nan"))
            p, t = torch.softmax(torch.cat(self._pred), -1), torch.cat(self._target)
            aucs = []
            for c in range(self.num_classes):
                pos, neg = p[t == c, c], p[t != c, c]
                if len(pos) == 0 or len(neg) == 0:
                    continue
                gt = (pos[:, None] > neg[None, :]).float().mean()
                eq = (pos[:, None] == neg[None, :]).float().mean()
                aucs.append(gt + 0.5 * eq)
            return torch.stack(aucs).mean() if aucs else torch.tensor(float("nan"))


class VeryBasicModel(_LightningBase):
    """reference base_model.py:10-81"""

    def __init__(self, save_hyperparameters=True):
        super().__init__()
        if save_hyperparameters:
            self.save_hyperparameters()
        self._step_train = -1
        self._step_val = -1
        self._step_test = -1

    def forward(self, x, cond=None):
        raise NotImplementedError

    def _step(self, batch: dict, batch_idx: int, state: str, step: int):
        raise NotImplementedError

    def _epoch_end(self, state: str):
        return

    def training_step(self, batch: dict, batch_idx: int):
        self._step_train += 1
        return self._step(batch, batch_idx, "train", self._step_train)

    def validation_step(self, batch: dict, batch_idx: int):
        self._step_val += 1
        return self._step(batch, batch_idx, "val", self._step_val)

    def test_step(self, batch: dict, batch_idx: int):
        self._step_test += 1
        return self._step(batch, batch_idx, "test", self._step_test)

    def on_train_epoch_end(self) -> None:
        self._epoch_end("train")

    def on_validation_epoch_end(self) -> None:
        self._epoch_end("val")

    def on_test_epoch_end(self, outputs=None) -> None:
        self._epoch_end("test")

    @classmethod
    def save_best_checkpoint(cls, path_checkpoint_dir, best_model_path):
        with open(Path(path_checkpoint_dir) / "best_checkpoint.json", "w") as f:
            json.dump({"best_model_epoch": Path(best_model_path).name}, f)

    @classmethod
    def _get_best_checkpoint_path(cls, path_checkpoint_dir, **kwargs):
        with open(Path(path_checkpoint_dir) / "best_checkpoint.json", "r") as f:
            rel = Path(json.load(f)["best_model_epoch"])
        return Path(path_checkpoint_dir) / rel

    @classmethod
    def load_best_checkpoint(cls, path_checkpoint_dir, **kwargs):
        return cls.load_from_checkpoint(cls._get_best_checkpoint_path(path_checkpoint_dir), **kwargs)

    def load_pretrained(self, checkpoint_path, map_location=None, **kwargs):
        checkpoint_path = Path(checkpoint_path)
        if checkpoint_path.is_dir():
            checkpoint_path = self._get_best_checkpoint_path(checkpoint_path, **kwargs)
        checkpoint = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        return self.load_weights(checkpoint["state_dict"], **kwargs)

    def load_weights(self, pretrained_weights, strict=True, **kwargs):
        keep = kwargs.get("filter", lambda key: key in pretrained_weights)
        weights = self.state_dict()
        weights.update({k: v for k, v in pretrained_weights.items() if keep(k)})
        self.load_state_dict(weights, strict=strict)
        return self


class BasicModel(VeryBasicModel):
    """reference base_model.py:86-110"""

    def __init__(self, optimizer=torch.optim.Adam, optimizer_kwargs={"lr": 1e-3, "weight_decay": 1e-2},
                 lr_scheduler=None, lr_scheduler_kwargs={}, save_hyperparameters=True):
        super().__init__(save_hyperparameters=save_hyperparameters)
        if save_hyperparameters:
            self.save_hyperparameters()
        self.optimizer = optimizer
        self.optimizer_kwargs = optimizer_kwargs
        self.lr_scheduler = lr_scheduler
        self.lr_scheduler_kwargs = lr_scheduler_kwargs

    def configure_optimizers(self):
        optimizer = self.optimizer(self.parameters(), **self.optimizer_kwargs)
        if self.lr_scheduler is None:
            return [optimizer]
        scheduler = self.lr_scheduler(optimizer, **self.lr_scheduler_kwargs)
        return [optimizer], [{"scheduler": scheduler, "interval": "step", "frequency": 1}]


class BasicClassifier(BasicModel):
    """reference base_model.py:116-181: CE loss + accuracy / AUROC per split."""

    def __init__(self, in_ch, out_ch, spatial_dims, loss=torch.nn.CrossEntropyLoss, loss_kwargs={},
                 optimizer=torch.optim.AdamW, optimizer_kwargs={"lr": 1e-4, "weight_decay": 1e-2},
                 lr_scheduler=None, lr_scheduler_kwargs={}, aucroc_kwargs={"task": "multiclass"},
                 acc_kwargs={"task": "multiclass"}, save_hyperparameters=True):
        super().__init__(optimizer, optimizer_kwargs, lr_scheduler, lr_scheduler_kwargs, save_hyperparameters)
        self.in_ch = in_ch
        self.out_ch = out_ch
        self.spatial_dims = spatial_dims
        self.loss_func = loss(**loss_kwargs)
        self.loss_kwargs = loss_kwargs
        aucroc_kwargs = dict(aucroc_kwargs, num_classes=out_ch)
        acc_kwargs = dict(acc_kwargs, num_classes=out_ch)
        # 'train' is not allowed as a ModuleDict key, hence the trailing underscore (reference l.144-145)
        self.auc_roc = nn.ModuleDict({s: AUROC(**aucroc_kwargs) for s in ["train_", "val_", "test_"]})
        self.acc = nn.ModuleDict({s: Accuracy(**acc_kwargs) for s in ["train_", "val_", "test_"]})

    def _step(self, batch: dict, batch_idx: int, state: str, step: int):
        target = batch["target"]
        self.batch_size = batch_size = target.shape[0]
        pred = self(**batch)  # the whole batch dict is splatted (reference l.155)
        loss = self.compute_loss(pred, target.to(pred.device))
        with torch.no_grad():
            self.acc[state + "_"].update(pred, target)
            self.auc_roc[state + "_"].update(pred, target)
            self.log(f"{state}/loss", loss, batch_size=batch_size, on_step=True, on_epoch=True, sync_dist=False)
        return loss

    def _epoch_end(self, state):
        for name, metric in (("ACC", self.acc[state + "_"]), ("AUC_ROC", self.auc_roc[state + "_"])):
            self.log(f"{state}/{name}", metric.compute(), batch_size=getattr(self, "batch_size", 1), on_step=False,
                     on_epoch=True, sync_dist=True)
            metric.reset()

    def compute_loss(self, pred, target):
        return self.loss_func(pred, target)

"""Glue between the nn.Module parameter tree and the native engine: flat parameter storage + autograd bridge."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .engine import LiftEngine


class _LiftFunction(torch.autograd.Function):
    """(x, *params) -> poses[, scores] through mp_model_forward; backward through mp_model_backward.

    The engine keeps the activations of ONE forward (its arena is sized for one batch).  Ordinary autograd semantics are kept on top of that:
    when a backward arrives for a forward whose activations have been overwritten since (``l1 = f(model(x1)); l2 = f(model(x2));
    (l1 + l2).backward()``), that forward is re-run first from the saved input with the same DropPath stream position - the engine is
    deterministic (same bits), so the gradients are those of the original graph, at the price of one extra forward.  The reference's own
    loop (one forward, one backward) never pays it."""

    @staticmethod
    def forward(ctx, model, x, train, masks, infer, *params):
        eng = model._engine
        model._step_counter += 1
        poses, scores = eng.forward(model._flat, x, train=train, masks=masks, seed=model._seed, step=model._step_counter, infer=infer)
        ctx.held = [id(eng), eng.forward_serial if not infer else -1]            # the engine forward whose activations belong to this graph node
        ctx.model = model
        ctx.save_for_backward(x) if masks is None else ctx.save_for_backward(x, masks)   # x is read again by the embedding backward
        ctx.has_scores = scores is not None
        ctx.fwd = (model._step_counter, bool(train), model._seed)
        ctx.param_token = model._param_token()                                   # the parameter bits this graph node was computed from
        if scores is None:
            return poses
        return poses, scores

    @staticmethod
    def backward(ctx, d_poses, d_scores=None):
        model = ctx.model
        eng = model._engine
        step, train, seed = ctx.fwd
        if ctx.param_token != model._param_token():
            # (what autograd's saved-tensor version check says for an ordinary module: the engine reads the parameters again in the backward -
            # and in the re-run forward of the branch below - so gradients taken now would belong to a different graph)
            raise RuntimeError("manipose_amd: the model's parameters were modified in place (optimizer step, load_state_dict, ...) between this "
                               "forward and its backward; run the backward before updating the parameters")
        if eng is None or ctx.held != [id(eng), eng.forward_serial]:         # some other forward (of any caller) has run on the engine since
            x, *mk = ctx.saved_tensors
            if eng is None or eng.max_batch < x.shape[0]:
                raise RuntimeError("manipose_amd: the engine of this forward no longer exists (the model was moved or its batch capacity rebuilt)")
            eng.forward(model._flat, x, train=train, masks=mk[0] if mk else None, seed=seed, step=step, infer=False)
            ctx.held = [id(eng), eng.forward_serial]
        flat_grads = torch.zeros_like(model._flat)
        d_poses = d_poses.contiguous()
        d_scores = d_scores.contiguous() if (ctx.has_scores and d_scores is not None) else None
        eng.backward(model._flat, flat_grads, d_poses, d_scores)
        ctx.held = None                       # (a second backward through a retained graph re-runs the forward: the backward reuses scratch)
        model._last_flat_grad = flat_grads
        grads = tuple(flat_grads[off:off + n].view(p.shape) for (off, n), p in zip(model._slots, model._plist))
        return (None, None, None, None, None) + grads


class FusedLiftingMixin:
    """Mixed into ManifoldMixSTE / RMCLManifoldMixSTE: keeps every nn.Parameter a view of ONE flat device buffer laid
    out as the engine expects (mp_model_param_info), so that the engine, the fused Adam and the gradient all-reduce
    each see a single contiguous array while state-dict keys stay those of the reference."""

    def _init_fused(self, arch: str, cfg: dict):
        self._arch = arch
        self._engine_cfg = cfg
        self._engine: Optional[LiftEngine] = None
        self._flat: Optional[torch.Tensor] = None
        self._last_flat_grad: Optional[torch.Tensor] = None
        self._slots: List = []
        self._plist: List[torch.nn.Parameter] = []
        self._step_counter = 0
        self._seed = 42
        self._injected_masks: Optional[Dict[str, torch.Tensor]] = None
        self.precision = os.environ.get("MANIPOSE_PRECISION", "fp32")
        # engine options of THIS model (mp_model_config, ABI v7; read when the engine is (re)built): f16f8 0 / 1 / 2 / 3 and f16_backward = the
        # operand form of the qkv / fc1 (/ fc2) layers of a bf16x3 model (default: three bf16 products everywhere, bf16 backward);
        # side_stream / wgrad_stream = the engine's two extra HIP streams (off: everything on the caller's stream, same bits)
        self.f16f8 = 0
        self.f16_backward = False
        self.side_stream = True
        self.wgrad_stream = True
        self.hazard_check = False         # mp_model_config::debug bit 0: the engine's host-side stream-hazard check (engine.hazard_report())
        self.max_batch_hint = 0

    # -- engine / flat storage -----------------------------------------------------------------
    def _engine_options(self) -> dict:
        return dict(f16f8=int(self.f16f8), f16_backward=bool(self.f16_backward), side_stream=bool(self.side_stream), wgrad_stream=bool(self.wgrad_stream),
                    hazard_check=bool(self.hazard_check))

    def _ensure_engine(self, B: int, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("manipose_amd: this model runs on MI355X through hand-written HIP kernels only; move the "
                               "model and its inputs to a ROCm device (model.cuda()). There is no CPU fallback.")
        eng = self._engine
        if eng is None or eng.max_batch < B or eng.device != device:
            with torch.cuda.device(device):
                self._engine = None
                eng = LiftEngine(arch=self._arch, max_batch=max(B, self.max_batch_hint), precision=self.precision,
                                 **self._engine_cfg, **self._scale_cfg(), **self._engine_options())
            self._engine = eng
            self._flat = None
        if self._flat is None or not self._views_intact():
            self._flatten(eng, device)

    def _scale_cfg(self) -> dict:
        """Attention / residual scales and MuReadout input multipliers of the two backbones as the engine takes them (0 = default):
        explicit qk_scale arguments and the muP mode of the reference (mix_ste.py:243,330; mup.MuReadout), read off the modules."""
        from ..mup_lite import MuReadout
        out = {}
        rot = getattr(self, "rotations_module", self)
        for tag, mod in (("rot", rot), ("seg", getattr(self, "segments_module", None))):
            if mod is None:
                continue
            blk = mod.STEblocks[0]
            d = mod.embed_dim // mod.num_heads
            out[f"qk_scale_{tag}"] = 0.0 if abs(blk.attn.scale - d ** -0.5) < 1e-12 else float(blk.attn.scale)
            out[f"resid_scale_{tag}"] = 0.0 if blk.residual_scale == 1.0 else float(blk.residual_scale)
            heads = [h.prediction_head for h in mod.head] if isinstance(mod.head, torch.nn.ModuleList) else [mod.head[1]]
            mults = {h.input_multiplier() if isinstance(h, MuReadout) else 1.0 for h in heads}
            if len(mults) != 1:
                raise NotImplementedError("manipose_amd: hypothesis heads with different MuReadout multipliers")
            if isinstance(mod.head, torch.nn.ModuleList):
                for h in mod.head:
                    if isinstance(h.score_head, MuReadout) and h.score_head.input_multiplier() != 1.0:
                        raise NotImplementedError("manipose_amd: a score head with a width multiplier (its fan-in, the joint count, is not a width)")
            m = mults.pop()
            out[f"readout_mult_{tag}"] = 0.0 if m == 1.0 else float(m)
        return out

    def _param_token(self):
        """Changes whenever the parameter bits may have: the autograd version counters of the flat buffer and of every parameter (each
        nn.Parameter is re-pointed at its slice through `.data`, so it counts its own in-place writes: load_state_dict, p.mul_(...)) and the
        count of fused optimizer steps (the Adam kernel writes through raw pointers).  ~300 integer reads per call."""
        return (self._flat._version if self._flat is not None else -1, sum(p._version for p in self._plist), getattr(self, "_fused_updates", 0))

    def _views_intact(self) -> bool:
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for (off, _), p in zip(self._slots, self._plist))

    def _flatten(self, eng: LiftEngine, device: torch.device):
        named = dict(self.named_parameters())
        if set(named) != {n for n, _, _ in eng.layout}:
            missing = {n for n, _, _ in eng.layout} ^ set(named)
            raise RuntimeError(f"manipose_amd: parameter tree does not match the engine layout: {sorted(missing)[:6]}")
        flat = torch.zeros(eng.flat_size, dtype=torch.float32, device=device)
        self._slots, self._plist = [], []
        for name, off, numel in eng.layout:
            p = named[name]
            if p.numel() != numel:
                raise RuntimeError(f"manipose_amd: {name} has {p.numel()} elements, engine expects {numel}")
            flat[off:off + numel].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[off:off + numel].view(p.shape)
            self._slots.append((off, numel))
            self._plist.append(p)
        self._flat = flat

    def flat_parameters(self) -> torch.Tensor:
        """The single contiguous fp32 buffer all parameters live in (engine layout); build it if needed.  On a CPU
        model this only lays the parameters out (layout-only engine handle): compute still requires a ROCm device."""
        p = next(self.parameters())
        if p.device.type != "cuda":
            if self._flat is None or self._flat.device != p.device or not self._views_intact():
                self._engine = LiftEngine(arch=self._arch, max_batch=0, precision=self.precision, **self._engine_cfg, **self._scale_cfg())
                self._flatten(self._engine, p.device)
            return self._flat
        if self._flat is None or self._flat.device != p.device or not self._views_intact():
            self._ensure_engine(max(1, self.max_batch_hint), p.device)
        return self._flat

    def flat_layout(self):
        """[(state-dict key, offset, numel)] of the flat buffer."""
        self.flat_parameters()
        return list(self._engine.layout)

    def set_droppath_masks(self, masks: Optional[Dict[str, torch.Tensor]]):
        """Inject DropPath multipliers (branch name -> per-sample tensor) for the next train-mode forwards
        (parity tests); ``None`` returns to the engine's own counter-based RNG."""
        self._injected_masks = masks

    def _run(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != self._engine_cfg["num_frame"] or x.shape[2] != self._engine_cfg["num_joints"] \
                or x.shape[3] != 2:
            raise AssertionError(f"expected input of shape (B, {self._engine_cfg['num_frame']}, "
                                 f"{self._engine_cfg['num_joints']}, 2), got {tuple(x.shape)}")
        self._ensure_engine(x.shape[0], x.device)
        x = x.contiguous().float()
        masks = None
        if self.training and self._injected_masks is not None:
            masks = self._engine.pack_masks(x.shape[0], self._injected_masks)
        # under torch.no_grad() (or with nothing to differentiate) no backward can follow: tell the engine
        infer = not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._plist)))
        return _LiftFunction.apply(self, x, self.training, masks, infer, *self._plist)

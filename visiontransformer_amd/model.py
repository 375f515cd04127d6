"""`ViTSegmentationModel` -- drop-in mirror of the reference class
(/root/reference/model/CE/classes.py:221-262; byte-identical copy at model/PAED/classes.py:372-413).

Same positional constructor, same `forward(x[B,3,H,W]) -> logits[B,C,H,W]`, same state-dict key
schema, same ValueErrors; the arithmetic runs in libvitseg.so (hand-written gfx950 kernels) on
the tensor's HIP stream.  PyTorch is storage only: one flat fp32 parameter arena (a single
nn.Parameter), a cached workspace tensor and the output tensor.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, params as _params
from .config import ViTSegConfig

_PRECISION = {"fp32": _lib.F32, "f32": _lib.F32, "float32": _lib.F32, "bf16": _lib.BF16, "bfloat16": _lib.BF16,
              "fp16": _lib.F16, "f16": _lib.F16, "float16": _lib.F16, "half": _lib.F16,
              "fp32x3": _lib.F32X3, "f32x3": _lib.F32X3}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _LogitsFn(torch.autograd.Function):
    """logits = forward(x) with libvitseg's backward: d loss / d logits -> d loss / d arena."""

    @staticmethod
    def forward(ctx, arena, model, x):
        ctx.model, ctx.x = model, x
        ctx.drop = model._next_dropout()
        return model._forward_train(x, want_logits=True, drop=ctx.drop)

    @staticmethod
    def backward(ctx, dlogits):
        grads, _ = ctx.model._backward(ctx.x, grad_logits=dlogits.to(torch.float32).contiguous(), drop=ctx.drop)
        return ctx.model._deliver_grad(grads), None, None


class _CELossFn(torch.autograd.Function):
    """Fused CE: forward + loss + backward in one go (no [B,C,S,S] logits tensor is returned).  With `grad_scale` the
    factor is folded into the CE gradient at its source (vitseg_backward's loss_scale) and the upstream gradient of the
    returned loss is taken to be 1: no arena-sized multiply.  The gradient is delivered by `_deliver_grad` (installed as
    `arena.grad` or added to it), not returned to autograd: AccumulateGrad would clone an arena-sized tensor it cannot
    steal (the model keeps a reference to its persistent buffer)."""

    @staticmethod
    def forward(ctx, arena, model, x, target, grad_scale):
        drop = model._next_dropout()
        model._forward_train(x, want_logits=False, drop=drop)
        grads, loss = model._backward(x, target=target, drop=drop, loss_scale=1.0 if grad_scale is None else grad_scale)
        ctx.grads, ctx.model = grads, model
        ctx.prescaled = grad_scale is not None
        return loss

    @staticmethod
    def backward(ctx, dloss):
        grads, ctx.grads = ctx.grads, None
        if not ctx.prescaled:
            grads.mul_(dloss)   # in place: the buffer is ours until it is delivered
        return ctx.model._deliver_grad(grads), None, None, None, None


class ViTSegmentationModel(nn.Module):
    def __init__(self, num_classes, patch_size, hidden_size, num_hidden_layers, num_attention_heads, *,
                 image_size: int = 224, intermediate_size: int = 3072, precision: str = "fp32",
                 dropout: float = 0.1, device=None):
        super().__init__()
        self.cfg = ViTSegConfig(num_classes, patch_size, hidden_size, num_hidden_layers, num_attention_heads,
                                image_size=image_size, intermediate_size=intermediate_size)
        self.precision = _PRECISION[precision]
        # hidden_dropout_prob = attention_probs_dropout_prob = 0.1 in the reference (classes.py:233-234); active only
        # in train() mode with autograd on, like nn.Dropout.  `dropout_seed` + a step counter select the masks.
        self.dropout = float(dropout)
        self.dropout_seed = 0x5EED
        self._dropout_step = 0
        # data-parallel gradient exchange: "overlap" = bucketed all-reduce behind events inside the backward,
        # "after" = one flat all-reduce by the caller (dist.allreduce_grads), see dist.sync_grads
        self.grad_sync = "overlap"
        self.grad_bucket_mb = 48.0
        self._buckets = None
        self._grads_reduced = False
        self._require_sync = True
        self._graphs = {}   # (batch, with logits) -> captured hipGraph of the forward (predict_mask_graphed)
        n = _lib.param_count(self.cfg)  # validates the configuration (ValueError on unsupported shapes)
        self.arena = nn.Parameter(torch.zeros(n, dtype=torch.float32, device=device))
        self._views: Optional[Dict[str, torch.Tensor]] = None
        self._views_key = None
        self._ws = {}
        self._arena_bf16 = None
        self._bf16_version = None
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def named_views(self) -> Dict[str, torch.Tensor]:
        """reference parameter name -> live view into the arena."""
        key = (self.arena.data_ptr(), self.arena.device)
        if self._views is None or self._views_key != key:
            self._views = _params.arena_views(self.cfg, self.arena.data)
            self._views_key = key
        return self._views

    @torch.no_grad()
    def reset_parameters(self, seed: int = 0):
        """Initialisers of the reference: HF ViT init (N(0, 0.02) weights, zero biases, LayerNorm 1/0,
        trunc-normal cls/pos; modeling_vit.py:324-332) and torch's Conv2d default for seg_head."""
        g = torch.Generator().manual_seed(seed)
        self.arena.zero_()
        for name, v in self.named_views().items():
            if name.startswith("seg_head."):
                w = self.named_views()[name.rsplit(".", 1)[0] + ".weight"]
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
                bound = 1.0 / fan_in ** 0.5
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
            elif "layernorm" in name:
                v.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith(".bias"):
                v.zero_()
            elif name.endswith("cls_token") or name.endswith("position_embeddings"):
                v.copy_(torch.nn.init.trunc_normal_(torch.empty(v.shape), std=0.02, generator=g))
            else:
                v.copy_(torch.randn(v.shape, generator=g) * 0.02)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = OrderedDict() if destination is None else destination
        for k, v in self.named_views().items():
            out[prefix + k] = v if keep_vars else v.detach().clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        views = self.named_views()
        seen, unexpected = set(), []
        for k, v in state_dict.items():
            ck = _params.canonical_key(k)
            if ck.startswith("backbone.pooler."):
                continue  # computed then discarded by the reference (modeling_vit.py:386)
            if ck not in views:
                unexpected.append(k)
                continue
            if tuple(v.shape) != tuple(views[ck].shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model "
                                   f"{tuple(views[ck].shape)}")
            views[ck].copy_(torch.as_tensor(v).to(views[ck].device, torch.float32))
            seen.add(ck)
        missing = [k for k in views if k not in seen]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        self._bf16_version = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._views = None
        self._ws.clear()
        self._arena_bf16 = None
        self._bf16_version = None
        return r

    # ------------------------------------------------------------------ launch plumbing
    def _check_input(self, x: torch.Tensor):
        cfg = self.cfg
        if x.dim() != 4:
            raise ValueError(f"expected a [B, C, H, W] tensor, got shape {tuple(x.shape)}")
        if x.shape[1] != cfg.num_channels:  # modeling_vit.py:63-68
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             f"configuration. Expected {cfg.num_channels} but got {x.shape[1]}.")
        if x.shape[2] != cfg.image_size or x.shape[3] != cfg.image_size:  # modeling_vit.py:152-156
            raise ValueError(f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model "
                             f"({cfg.image_size}*{cfg.image_size}).")
        if not x.is_cuda or x.device != self.arena.device:
            raise RuntimeError("ViTSegmentationModel runs on the MI355X only: move the model and the input to the "
                               f"same HIP device (input on {x.device}, parameters on {self.arena.device}). "
                               "There is no CPU fallback.")

    def forward_route(self, batch: int) -> str:
        """"small" / "large": the kernel family an inference forward of this batch size runs on (results are bit-identical for
        every batch size inside one route)."""
        return _lib.forward_route(self.cfg, batch, self.precision)

    def workspace(self, batch: int) -> torch.Tensor:
        key = (batch, self.precision)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.query_workspace(self.cfg, batch, self.precision)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.arena.device)
            self._ws = {key: ws}  # keep one: a new batch size replaces the old workspace
        return ws

    def _bf16_arena(self):
        """Shadow of the arena in the operand format of `self.precision`: bf16 / IEEE half (2 bytes per value) or, for
        fp32x3, the pre-split (hi | lo halves) image with the fp32 arena's size and offsets.  Refreshed when the fp32
        master changes."""
        if self.precision == _lib.F32:
            return None
        ver = (self.arena._version, self.arena.data_ptr())
        if self._arena_bf16 is None or self._bf16_version != ver:
            dt, cast = {_lib.BF16: (torch.bfloat16, _lib.lib().vitseg_cast_params_bf16),
                        _lib.F16: (torch.float16, _lib.lib().vitseg_cast_params_f16),
                        _lib.F32X3: (torch.float32, _lib.lib().vitseg_cast_params_split)}[self.precision]
            if self._arena_bf16 is None:
                self._arena_bf16 = torch.empty(self.arena.numel(), dtype=dt, device=self.arena.device)
            _lib.check(cast(self.arena.data_ptr(), self._arena_bf16.data_ptr(), self.arena.numel(),
                            torch.cuda.current_stream().cuda_stream))
            self._bf16_version = ver
        return self._arena_bf16

    def _run(self, x: torch.Tensor, want_logits: bool, want_mask: bool, ws: Optional[torch.Tensor] = None):
        self._check_input(x)
        x = x.to(torch.float32).contiguous()  # modeling_vit.py:369-371 casts to the weight dtype
        B, S, Cc = x.shape[0], self.cfg.image_size, self.cfg.num_classes
        logits = torch.empty((B, Cc, S, S), dtype=torch.float32, device=x.device) if want_logits else None
        mask = torch.empty((B, S, S), dtype=torch.uint8, device=x.device) if want_mask else None
        if ws is None:
            ws = self.workspace(B)
        lp = self._bf16_arena()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream().cuda_stream
            rc = _lib.lib().vitseg_forward(C.byref(_lib.CConfig.from_config(self.cfg)), self.arena.data_ptr(),
                                           _ptr(lp), x.data_ptr(), B, self.precision, _ptr(logits), _ptr(mask),
                                           ws.data_ptr(), ws.numel(), stream)
        _lib.check(rc)
        return logits, mask

    # ------------------------------------------------------------------ training plumbing
    def _train_workspace(self, batch: int) -> torch.Tensor:
        key = ("train", batch)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(_lib.train_workspace(self.cfg, batch, self.precision), dtype=torch.uint8,
                             device=self.arena.device)
            self._ws = {key: ws}
        return ws

    def _next_dropout(self):
        """(p, seed) of the next training forward; p = 0 outside train() mode."""
        if not self.training or self.dropout <= 0.0:
            return (0.0, 0)
        self._dropout_step += 1
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        return (self.dropout, (self.dropout_seed * 0x9E3779B97F4A7C15 + self._dropout_step * 0x100000001B3 + rank) & (2 ** 64 - 1))

    def _forward_train(self, x: torch.Tensor, want_logits: bool, drop=(0.0, 0)):
        self._check_input(x)
        x = x.to(torch.float32).contiguous()
        B, S = x.shape[0], self.cfg.image_size
        ws = self._train_workspace(B)
        logits = torch.empty((B, self.cfg.num_classes, S, S), dtype=torch.float32, device=x.device) \
            if want_logits else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().vitseg_forward_train(
                C.byref(_lib.CConfig.from_config(self.cfg)), self.arena.data_ptr(), _ptr(self._bf16_arena()),
                x.data_ptr(), B, self.precision, drop[0], drop[1], _ptr(logits), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream().cuda_stream))
        return logits

    def no_sync(self):
        """Context manager for gradient accumulation (what DDP's `no_sync` is for): backwards inside it only accumulate
        locally; the all-reduce happens once, on the first backward outside (`dist.sync_grads` after it)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._require_sync = self._require_sync, False
            try:
                yield
            finally:
                self._require_sync = old
        return ctx()

    def _overlap_active(self) -> bool:
        """Bucketed all-reduce inside the backward: "overlap" = whenever a process group with > 1 rank exists,
        "force" = also with one rank (tests), "after" = never (callers use dist.allreduce_grads).  Never inside
        `no_sync()`, and never while a locally accumulated gradient is pending (`arena.grad` set): the kernels write
        THIS micro-batch's gradient, the sum over micro-batches only exists after autograd has accumulated it, so that
        step is reduced once, flat, by `dist.sync_grads`."""
        import torch.distributed as td
        if not self._require_sync or self.arena.grad is not None:
            return False
        if self.grad_sync == "force":
            return td.is_available() and td.is_initialized()
        return self.grad_sync == "overlap" and td.is_available() and td.is_initialized() and td.get_world_size() > 1

    def _bucket_state(self):
        if self._buckets is None:
            from .dist import BucketReducer
            ranges = _lib.grad_buckets(self.cfg)
            events = [torch.cuda.Event() for _ in ranges]
            for e in events:
                e.record()  # instantiates the hipEvent_t so its handle can cross the C ABI
            handles = (C.c_void_p * len(events))(*[e.cuda_event for e in events])
            self._buckets = (BucketReducer(ranges, self.grad_bucket_mb), events, handles, torch.cuda.Stream())
        return self._buckets

    def _take_grad_buffer(self) -> torch.Tensor:
        """The arena-sized buffer vitseg_backward writes: ONE persistent tensor, reused step after step.  It cannot be
        reused while it is still somebody's gradient: handed to an autograd node whose backward has not run yet
        (`_grad_busy`, cleared by `_release_grad_buffer`), or installed as `arena.grad` (accumulation pending, or
        `zero_grad(set_to_none=False)`); then this call gets a temporary of its own."""
        buf = getattr(self, "_grad_buf", None)
        pending = self.arena.grad
        usable = (buf is not None and buf.shape == self.arena.shape and buf.device == self.arena.device
                  and not getattr(self, "_grad_busy", False)
                  and (pending is None or pending.data_ptr() != buf.data_ptr()))
        if usable:
            self._grad_busy = True
            return buf
        fresh = torch.empty_like(self.arena.data)
        if buf is None or buf.shape != self.arena.shape or buf.device != self.arena.device:
            self._grad_buf, self._grad_busy = fresh, True
        return fresh

    def _release_grad_buffer(self, grads: torch.Tensor) -> None:
        if getattr(self, "_grad_buf", None) is not None and grads.data_ptr() == self._grad_buf.data_ptr():
            self._grad_busy = False

    def _deliver_grad(self, grads: torch.Tensor) -> None:
        """Hands d loss / d arena to the parameter from inside an autograd backward and returns None for autograd (= no
        gradient through the graph edge): with no gradient pending, the buffer vitseg_backward wrote BECOMES `arena.grad`
        -- no copy; `_take_grad_buffer` will not hand it out again while it is installed -- otherwise it is added to the
        pending one (gradient accumulation), after which the buffer is free again."""
        with torch.no_grad():
            if self.arena.grad is None:
                self.arena.grad = grads
            else:
                self.arena.grad.add_(grads)
        self._release_grad_buffer(grads)
        return None

    def _backward(self, x: torch.Tensor, target: Optional[torch.Tensor] = None,
                  grad_logits: Optional[torch.Tensor] = None, drop=(0.0, 0), loss_scale: float = 1.0):
        x = x.to(torch.float32).contiguous()
        B = x.shape[0]
        ws = self._train_workspace(B)
        grads = self._take_grad_buffer()
        loss = torch.zeros((), dtype=torch.float32, device=x.device) if target is not None else None
        overlap = self._overlap_active()
        with torch.cuda.device(x.device):
            reducer, events, handles, comm = self._bucket_state() if overlap else (None, None, None, None)
            _lib.check(_lib.lib().vitseg_backward(
                C.byref(_lib.CConfig.from_config(self.cfg)), self.arena.data_ptr(), _ptr(self._bf16_arena()),
                x.data_ptr(), B, self.precision, drop[0], drop[1],
                _ptr(target), int(target is not None and target.dtype == torch.uint8), _ptr(grad_logits),
                grads.data_ptr(), _ptr(loss), float(loss_scale), handles, ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream().cuda_stream))
            if overlap:
                # the whole backward is enqueued by now; each ring starts when its bucket's event fires and the
                # compute stream only rejoins after the last one (what DDP's finalize does)
                for w in reducer.reduce(grads, events, comm):
                    w.wait()
                self._grads_reduced = True
        return grads, loss

    def _needs_grad(self) -> bool:
        return torch.is_grad_enabled() and self.arena.requires_grad

    # ------------------------------------------------------------------ reference surface
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """logits [B, C, H, W] (model/CE/classes.py:246-262).  Differentiable w.r.t. the parameters when
        autograd is on (the backward runs in libvitseg, see vitseg_backward).  In train() mode dropout
        (`self.dropout`, reference 0.1) is applied at the four HF sites with a counter-based generator."""
        if self._needs_grad():
            return _LogitsFn.apply(self.arena, self, x)
        logits, _ = self._run(x, True, False)
        return logits

    @torch.no_grad()
    def predict_mask(self, x: torch.Tensor, return_logits: bool = False):
        """uint8 [B, H, W] = argmax_c sigmoid(logits) with first-index ties, i.e. the reference scripts'
        `logits.sigmoid()` + `argmax` (model/CE/testViTModel.py:122-126), fused into the decoder tail."""
        logits, mask = self._run(x, return_logits, True)
        return (mask, logits) if return_logits else mask

    @torch.no_grad()
    def predict_mask_graphed(self, x: torch.Tensor, return_logits: bool = False):
        """`predict_mask` replayed from a captured hipGraph (one per batch size and output set): the ~110 kernel
        launches of a forward become one graph launch.  Measured (tools/latency_probe.py, ViT-B/16): no gain at batch
        1-8 -- 1.4 ms (bf16) to 5.9 ms (fp32) per forward at 224x224 is GPU time of under-filled GEMM launches, not host
        launch time -- so nothing uses it by default; it is there for smaller models / faster hosts, and bit-identical
        to the eager path.  Inputs are copied into the graph's static buffer; the returned tensors are the graph's
        static outputs and are overwritten by the next call with the same batch size (clone them to keep them).
        Re-captured automatically when the parameters change."""
        self._check_input(x)
        key = (int(x.shape[0]), bool(return_logits))
        ver = (self.arena._version, self.arena.data_ptr())
        g = self._graphs.get(key)
        if g is None or g["ver"] != ver:
            xs = x.to(torch.float32).contiguous().clone()
            ws = torch.empty(_lib.query_workspace(self.cfg, key[0], self.precision), dtype=torch.uint8,
                             device=self.arena.device)   # owned by the graph: `workspace()` recycles its buffer
            side = torch.cuda.Stream(device=xs.device)
            side.wait_stream(torch.cuda.current_stream(xs.device))
            with torch.cuda.stream(side):            # warm-up outside the capture: workspace, shadow arena, attributes
                for _ in range(2):
                    self._run(xs, return_logits, True, ws)
            torch.cuda.current_stream(xs.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph(keep_graph=True)   # the hipGraph_t stays readable (graph_nodes)
            with torch.cuda.graph(graph):
                logits, mask = self._run(xs, return_logits, True, ws)
            graph.instantiate()
            g = dict(graph=graph, x=xs, ws=ws, logits=logits, mask=mask, ver=ver)
            self._graphs[key] = g
        g["x"].copy_(x, non_blocking=True)
        g["graph"].replay()
        return (g["mask"], g["logits"]) if return_logits else g["mask"]

    def graph_nodes(self, batch: int, return_logits: bool = False) -> Optional[int]:
        """Launches per forward = nodes of the hipGraph `predict_mask_graphed` captured for this batch size (None if it
        has not been captured yet)."""
        g = self._graphs.get((int(batch), bool(return_logits)))
        if g is None:
            return None
        hip = C.CDLL("libamdhip64.so")
        hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        n = C.c_size_t()
        rc = hip.hipGraphGetNodes(C.c_void_p(g["graph"].raw_cuda_graph()), None, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"hipGraphGetNodes failed with {rc}")
        return int(n.value)

    @torch.no_grad()
    def predict_mask_tiled(self, x: torch.Tensor) -> torch.Tensor:
        """Masks for images LARGER than the model's image size by spatial tiling (BASELINE configs[4]: 1024x1024
        inputs through an image_size=512 model as four 512x512 tiles per image; build-defined).
        x: [B, 3, H, W] with H, W multiples of image_size -> uint8 [B, H, W].  Tiles are independent units, so they
        simply extend the batch; each tile sees only its own pixels (no cross-tile attention)."""
        S = self.cfg.image_size
        B, Cc, H, W = x.shape
        if H % S or W % S:
            raise ValueError(f"tiled prediction needs H and W to be multiples of {S}, got {H}x{W}")
        ty, tx = H // S, W // S
        tiles = x.reshape(B, Cc, ty, S, tx, S).permute(0, 2, 4, 1, 3, 5).reshape(B * ty * tx, Cc, S, S).contiguous()
        m = self.predict_mask(tiles)
        return m.reshape(B, ty, tx, S, S).permute(0, 1, 3, 2, 4).reshape(B, H, W).contiguous()

    def ce_loss(self, x: torch.Tensor, target: torch.Tensor, grad_scale: Optional[float] = None) -> torch.Tensor:
        """`nn.CrossEntropyLoss()(self(x), target)` (model/CE/classes.py:268,280) as a device scalar, without
        materialising the [B, C, S, S] logits: forward to the low-res map, then the fused upsample+CE kernel.
        `target`: class indices [B, S, S], torch.long (reference) or torch.uint8, on the model's device.
        `grad_scale` (optional): the gradient `loss.backward()` deposits is grad_scale * d loss / d params, folded into
        the CE gradient inside the kernel (e.g. 1 / accumulate_grad_batches); call `.backward()` on the returned loss
        itself then -- an upstream factor is ignored in this mode."""
        S, B = self.cfg.image_size, x.shape[0]
        if tuple(target.shape) != (B, S, S) or target.dtype not in (torch.int64, torch.uint8):
            raise ValueError(f"target must be int64/uint8 [B, {S}, {S}], got {target.dtype} {tuple(target.shape)}")
        if self._needs_grad():
            return _CELossFn.apply(self.arena, self, x, target.to(self.arena.device).contiguous(), grad_scale)
        with torch.no_grad():
            _, _ = self._run(x, False, True)  # fills the low-res logits (mask output is a by-product)
            low = self.debug_buffer(B, _lib.BUF_LOWRES)
            target = target.to(self.arena.device).contiguous()
            scratch = torch.empty(_lib.lib().vitseg_ce_scratch_bytes(B, S), dtype=torch.uint8, device=low.device)
            loss = torch.empty((), dtype=torch.float32, device=low.device)
            _lib.check(_lib.lib().vitseg_ce_loss(low.data_ptr(), target.data_ptr(), int(target.dtype == torch.uint8),
                                                 None, scratch.data_ptr(), loss.data_ptr(), B, self.cfg.num_classes,
                                                 self.cfg.grid, S, torch.cuda.current_stream().cuda_stream))
        return loss

    @torch.no_grad()
    def debug_buffer(self, batch: int, which: int) -> torch.Tensor:
        """fp32 view of a workspace buffer defined after forward (parity tests)."""
        off, n = _lib.workspace_offset(self.cfg, batch, self.precision, which)
        return self.workspace(batch)[off:off + n].view(torch.float32)

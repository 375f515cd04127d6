#!/usr/bin/env python3
"""Benchmark of the ViT-segmentation hot path on MI355X (contract: see the task prompt).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one per-GPU batch of synthetic 512x512 images that
are already resident in HBM: ViTSegmentationModel forward -> fp32 logits [B,C,512,512] AND the
uint8 sigmoid->argmax mask.  Workload at N=1 = BASELINE.json configs[1]: ViT-B/16 seg inference,
batch 32 x 512x512, fp32.  N>1: one process per GPU, every rank runs the same per-GPU batch on
its own shard of the global image stream (weak scaling, images are independent units; no
data-path collective -- SURVEY.md section 8e); value = all ranks' images / max-over-ranks time.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel: hipEvent time measured inside the
timed steps on the launch stream, algorithmic FLOPs per launch) and `cpu_baseline` (the oracle,
a CPU port of the reference path, on a bounded sample on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from visiontransformer_amd import _lib, synth  # noqa: E402
from visiontransformer_amd.config import vit_base16  # noqa: E402
from visiontransformer_amd.model import ViTSegmentationModel  # noqa: E402

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "f32x3": 2500.0 / 3}  # x3: 3 fp16 MFMAs per product  # dense MFMA peaks, MI355X_MICROARCH.md
KERNEL_NAMES = {  # precision -> kind -> kernel symbol as rocprofv3 prints it
    # fp32 linears with >= 128 tiles of 256x128: the persistent kernel gemm_f32p_kernel<EPI, DROP, AUX, INL> (csrc/gemm_f32p.hip)
    "f32": {"gemm_bias": "gemm_f32p_kernel<0, false, false, true>", "gemm_gelu": "gemm_f32p_kernel<1, false, false, true>",
            "gemm_resadd": "gemm_f32p_kernel<2, false, false, true>", "gemm_patch": "gemm_kernel<float, float, 1, 4, 0, 0, 0>",
            "gemm_conv3": "gemm_kernel<float, float, 2, 3, 0, 0, 0>",
            "attention": "attn_f32_kernel<false>"},   # (carries the CLS query too)
    # 16-bit operands: the 8-phase persistent kernel gemm_p8_kernel<T, OutT, EPI, TT> (csrc/gemm_p8.hip) whenever
    # M >= 2048, N % 256 == 0, K % 128 == 0 -- every encoder linear at the bench batch sizes
    # (the QKV projection -- bias epilogue, K <= 1024 -- runs on gemm_h16p_kernel<T>, csrc/gemm_h16p.hip)
    "bf16": {"gemm_bias": "gemm_h16p_kernel<unsigned short>",
             "gemm_gelu": "gemm_p8_kernel<unsigned short, unsigned short, 1, 0>",
             "gemm_resadd": "gemm_p8_kernel<unsigned short, float, 2, 0>",
             "gemm_patch": "gemm_kernel<float, float, 1, 4, 0, 0, 0>",
             "gemm_conv3": "gemm_bf16_large_kernel<unsigned short, float, 2, 3, 128>",
             "attention": "attn_bf16_kernel<false, false, false, unsigned short> + attn_cls_bf16_kernel<unsigned short>"},
    "f32x3": {"gemm_bias": "gemm_kernel<float, float, 0, 0, 0, 0, 2>", "gemm_gelu": "gemm_kernel<float, float, 0, 1, 0, 0, 2>",
              "gemm_resadd": "gemm_kernel<float, float, 0, 2, 0, 0, 2>", "gemm_patch": "gemm_kernel<float, float, 1, 4, 0, 0, 2>",
              "gemm_conv3": "gemm_kernel<float, float, 2, 3, 0, 0, 2>",
              "attention": "attn_x3_kernel<false> + attn_cls_f32_kernel"},
    # rocprofv3's demangler does not know _Float16 (DF16_): the IEEE-half instantiations appear mangled in its CSVs
    "f16": {"gemm_bias": "_ZN6vitseg12_GLOBAL__N_116gemm_h16p_kernelIDF16_EEvNS_8GemmArgsE",
            "gemm_gelu": "_ZN6vitseg14gemm_p8_kernelIDF16_DF16_Li1ELi0EEEvNS_8GemmArgsE",
            "gemm_resadd": "_ZN6vitseg14gemm_p8_kernelIDF16_fLi2ELi0EEEvNS_8GemmArgsE",
            "gemm_patch": "gemm_kernel<float, float, 1, 4, 0, 0, 0>",
            "gemm_conv3": "_ZN6vitseg22gemm_bf16_large_kernelIDF16_fLi2ELi3ELi128EEEvNS_8GemmArgsE",
            "attention": "_ZN6vitseg12_GLOBAL__N_116attn_bf16_kernelILb0ELb0ELb0EDF16_EEvPKtPtPfiiiNS_8DropArgsEPKj + "
                         "_ZN6vitseg12_GLOBAL__N_120attn_cls_bf16_kernelIDF16_EEvPKtPtPfiiiNS_8DropArgsE"},
    # training step (--mode train): kernel groups bracketed by the VITSEG_K_TRAIN_* scopes (csrc/vitseg_train.hip)
    "train_bf16": {"train_gemm_fwd": "gemm_h16p_kernel<unsigned short> (QKV) + gemm_p8_kernel<unsigned short, *, {1,2}, 0> (the four forward linears)",
                   "train_dgrad": "gemm_p8_kernel<unsigned short, *, {0,5}, 0> + gemm_h16p_kernel<unsigned short> (o_proj) + transpose_bf16_kernel (activation gradients)",
                   "train_wgrad": "gemm_p8_kernel<unsigned short, float, 0, 1> + splitk_reduce_kernel (weight gradients)",
                   "train_attn_fwd": "attn_dropmask_kernel + attn_bf16_kernel<true, false, true, unsigned short> + attn_cls_bf16_kernel<unsigned short>",
                   "train_attn_bwd": "attn_bwd_dq_bf16_kernel<true, false, true> (also forms delta) + attn_bwd_dkv_bf16_kernel<true, false, true> + "
                                     "attn_bwd_cls_finish_kernel<true>"},
    "train_f32": {"train_gemm_fwd": "gemm_f32p_kernel<{0,1,2}, ...> (the four forward linears)", "train_dgrad": "gemm_kernel<float, float, ...> (W T-form)",
                  "train_wgrad": "gemm_kernel<float, float, ...> (both T-form, split-K)",
                  "train_attn_fwd": "attn_f32_kernel<true>",
                  "train_attn_bwd": "attn_bwd_dq_f32_kernel + attn_bwd_dkv_f32_kernel"},
}


def algorithmic_bytes(precision, kind, cfg, batch):
    """Operand bytes one launch of the GEMM kind must move once (A + W + residual/bias + C), launch-averaged."""
    D, I = cfg.hidden_size, cfg.intermediate_size
    M = batch * ((cfg.image_size // cfg.patch_size) ** 2 + 1)
    e = 4 if precision == "f32" else 2
    shapes = {"gemm_bias": [(3 * D, D, e, 0)], "gemm_gelu": [(I, D, e, 0)],
              "gemm_resadd": [(D, D, 4, 4), (D, I, 4, 4)]}.get(kind)   # (N, K, bytes of C, bytes of residual)
    if not shapes:
        return None
    return round(sum(M * K * e + N * K * e + M * N * (c + r) + 4 * N for N, K, c, r in shapes) / len(shapes))


def pmc_traffic(precision, kernel_label, batch):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/collect_pmc.sh bench_<prec> ... ->
    tools/summarize_traffic.py -> profiles/rNN_traffic_<prec>.json, the newest round): counters cannot be read from inside this process,
    so the figure is the one rocprofv3 measured on this same command (batch 32).  None when no pass covers the run."""
    path = traffic_profile(precision)
    if batch != 32 or path is None:
        return None
    kernels = json.load(open(path))["kernels"]
    tot = n = 0.0
    for part in kernel_label.split(" + "):
        k = kernels.get(part.split(" (")[0].strip())
        if k is None:
            return None
        tot += k["hbm_bytes_per_launch"] * k["launches_sampled"]
        n += k["launches_sampled"]
    return round(tot / n) if n else None


def traffic_profile(precision):
    """Newest committed profiles/rNN_traffic_<precision>.json (None if there is none)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic_{precision}.json")))
    return found[-1] if found else None


def cpu_baseline(cfg, sd_np, images_np, gpu_logits, gpu_mask, seconds_budget=25.0, n_images=2):
    """Times the oracle (CPU port of the reference model/CE path) on the first images of the batch
    and uses the same run as the live parity check of the GPU output."""
    from oracle import vitseg_oracle as O
    # the GPU box gives one-GPU jobs a 16-core share of the host; more threads than that oversubscribe
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("VITSEG_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    n = min(n_images, images_np.shape[0])
    x = torch.from_numpy(images_np[:n])
    with torch.no_grad():
        t0 = time.perf_counter()
        logits = O.forward(x, sd, cfg)  # warm-up + parity reference
        mask = O.predict_mask(logits)
        first = time.perf_counter() - t0
        times = []
        while sum(times) + first < seconds_budget and len(times) < 5:
            t0 = time.perf_counter()
            mask = O.predict_mask(O.forward(x, sd, cfg))
            times.append(time.perf_counter() - t0)
    t = float(np.median(times)) if times else first
    base = {"value": n / t, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (pure-torch restatement of model/CE forward + sigmoid/argmax), fp32, "
                      f"{n} x {cfg.image_size}x{cfg.image_size} images, median of {max(len(times), 1)} runs"}
    return base, parity_vs(gpu_logits, gpu_mask, logits, mask), (logits, mask)


def parity_vs(gpu_logits, gpu_mask, logits, mask):
    """GPU output of the first images against oracle logits / mask of the same images.  `mask_mismatch_at_stable_pixels` must be
    0: a stable pixel is one whose sigmoid -> first-max decision no logit move of 2 x the measured error can change
    (oracle mask_stable, the criterion of tests/test_gpu_forward.py); `unstable_pixels` is the share that criterion exempts."""
    from oracle import vitseg_oracle as O
    n = logits.shape[0]
    err = float((gpu_logits[:n].cpu() - logits).abs().max())
    differ = gpu_mask[:n].cpu().long() != mask
    stable = O.mask_stable(logits.float(), 2.0 * err + 1e-7)
    return {"logits_max_abs_err": err, "mask_match": float((~differ).float().mean()),
            "mask_mismatch_at_stable_pixels": int((differ & stable).sum()),
            "unstable_pixels": float((~stable).float().mean()), "images_checked": n}


def run_tiled(precision, native, B, rank, dev, steps, warmup, barrier, procedural_weights=True):
    """ViT-L/16 on `B` 1024x1024 images per step, mask-only output.  Returns (cfg, seconds, profile, last mask)."""
    from visiontransformer_amd.config import vit_large16
    S = 1024 if native else 512
    cfg = vit_large16(num_classes=2, image_size=S)
    model = ViTSegmentationModel(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                                 cfg.num_attention_heads, image_size=S,
                                 precision={"f32": "fp32", "bf16": "bf16", "f16": "fp16"}[precision], device=dev).eval()
    if procedural_weights:   # (else: the constructor's random init, reference initialisers, torch generator seed 0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=1).items()})
    if native:
        x = torch.from_numpy(synth.make_images(cfg, B, seed=0, first_image=rank * B)).to(dev)
        run = model.predict_mask
    else:
        tiles = torch.from_numpy(synth.make_images(cfg, 4 * B, seed=0, first_image=rank * 4 * B)).to(dev)
        x = tiles.reshape(B, 2, 2, 3, 512, 512).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 1024, 1024).contiguous()
        del tiles
        run = model.predict_mask_tiled
    with torch.no_grad():
        for _ in range(warmup):
            mask = run(x)
        torch.cuda.synchronize()
        barrier()
        _lib.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            mask = run(x)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    return cfg, elapsed, prof, mask


def hbm_rooflines(prof, kinds):
    return {k: {"achieved_GBps": round(prof[k]["work"] / (prof[k]["ms"] * 1e-3) / 1e9, 1), "peak_GBps": 8000.0,
                "frac": round(prof[k]["work"] / (prof[k]["ms"] * 1e-3) / 8e12, 4),
                "bytes_per_launch": prof[k]["work"] / max(prof[k]["launches"], 1)}
            for k in kinds if prof[k]["ms"] > 0}


def bench_tiled(args, rank, world, dev, barrier):
    """BASELINE configs[4]: ViT-L/16 seg inference on 1024x1024 inputs, fp16 operands by default; `--batch` 1024^2 images
    per GPU (16 = the config's 128 over 8 GPUs).  Mask-only output, so the decoder tail writes 1 B/pixel.
    l16_1024_tiled: four 512x512 tiles through an image_size=512 model (build-defined tiling; N = 1025).
    l16_1024_native: the image as ONE sequence of N = 4097 tokens through an image_size=1024 model (SURVEY 8d's
    alternative: 3 738 GF per image, attention 44 % of it)."""
    native = args.workload == "l16_1024_native"
    B = args.batch
    cfg, elapsed, prof, mask = run_tiled(args.precision, native, B, rank, dev, args.steps, args.warmup, barrier)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops_img = (1.0 if native else 4.0) * cfg.forward_flops_per_image()
        peak = PEAK_TFLOPS[args.precision]
        mfma = {k: v for k, v in prof.items() if k.startswith("gemm") or k == "attention"}
        dom = max(mfma, key=lambda k: mfma[k]["ms"])
        d = mfma[dom]
        achieved = d["work"] / (d["ms"] * 1e-3) / 1e12
        print(json.dumps({
            "metric": "images/sec (1024x1024, one sequence of 4097 tokens) ViT-L/16 seg" if native
                      else "images/sec (1024x1024 as 4 tiles of 512x512) ViT-L/16 seg",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"ViT-L/16 seg inference, {B} x 1024x1024 per GPU "
                                   f"{'native (N = 4097 tokens)' if native else 'tiled 4 x 512x512'}, {args.precision}, "
                                   f"uint8 mask output (BASELINE.json configs[4])", "batch_per_gpu": B,
                       "global_batch": B * world, "tiles_per_step": (1 if native else 4) * B,
                       "mask_positive_fraction": float(mask.float().mean()),
                       "parallelism": f"batch-split x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None,
                         "kernel": KERNEL_NAMES[args.precision].get(dom, dom), "launches": d["launches"],
                         "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                         "flops_per_launch": d["work"] / max(d["launches"], 1)},
            "whole_model": {"flops_per_image": flops_img,
                            "achieved_tflops_per_gpu": round(value / world * flops_img / 1e12, 2),
                            "frac_of_peak": round(value / world * flops_img / 1e12 / peak, 4)},
            "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if not k.startswith("train_")},
            # the HBM-bandwidth-bound decoder head: low-res logits -> bilinear -> sigmoid/argmax -> 1 B/pixel
            "roofline_hbm": hbm_rooflines(prof, ("layernorm", "head1x1", "upsample"))}), flush=True)
    barrier()


def side_l16_tiled_f16(dev, steps=5):
    """BASELINE configs[4] at its per-GPU share inside the default line (rank 0, N = 1, outside the timed region):
    ViT-L/16, 16 images of 1024x1024 as 64 tiles of 512x512, fp16, uint8 mask output.  Random-init weights from the
    constructor (a throughput figure; the full-depth parity of this model is tests/test_gpu_forward.py)."""
    B = 16
    cfg, elapsed, prof, mask = run_tiled("f16", False, B, 0, dev, steps, 2, lambda: None, procedural_weights=False)
    peak = PEAK_TFLOPS["f16"]
    flops_img = 4.0 * cfg.forward_flops_per_image()
    value = B * steps / elapsed
    out = {"workload": "ViT-L/16 seg inference, 16 x 1024x1024 per GPU tiled 4 x 512x512, fp16, uint8 mask output "
                       "(BASELINE.json configs[4], one GPU's share of the batch of 128)",
           "images_per_s_per_gpu": round(value, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
           "whole_model_frac_of_f16_peak": round(value * flops_img / 1e12 / peak, 4),
           "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in prof.items()
                                  if not k.startswith("train_") and v["ms"] > 0},
           "roofline_hbm": hbm_rooflines(prof, ("layernorm", "head1x1", "upsample"))}
    del mask
    torch.cuda.empty_cache()
    return out


def bench_aux(args, rank, world, dev, barrier):
    """Rows f3 / f4 of SURVEY.md section 8 to the same bar as the hot path: --mode prep = Resize((512, 512)) + ToTensor
    of 3024x4032 RGB photos (uint8 in HBM -> fp32 NCHW), --mode eval = per-image class statistics of 512x512
    predictions against 256x256 label maps.  HBM-bound byte work: roofline = algorithmic bytes / hipEvent time."""
    from visiontransformer_amd.metrics import Evaluator
    from visiontransformer_amd.preprocess import Preprocessor
    B, S = args.batch, 512
    rs = np.random.RandomState(rank)
    if args.mode == "prep":
        H, W = 3024, 4032
        host = rs.randint(0, 256, size=(B, H, W, 3), dtype=np.uint8)
        src = torch.from_numpy(host).to(dev)
        pre = Preprocessor(S, dev)
        out = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
        run = lambda: pre.images(src, out)
        yb = pre._axis_tables(H, S)
        rows = yb[4] - yb[3]
        # read the photo once, write + re-read the uint8 intermediate, write the fp32 tensor
        algo = B * (H * W * 3 + 2 * rows * S * 3 + 3 * S * S * 4)
        what = f"Resize(({S},{S})) + ToTensor of {B} RGB photos {H}x{W} per GPU (uint8 resident in HBM)"
    else:
        pred = torch.from_numpy(rs.randint(0, 17, size=(B, S, S), dtype=np.uint8)).to(dev)
        gt = torch.from_numpy(rs.randint(0, 17, size=(B, 256, 256), dtype=np.uint8)).to(dev)
        ev = Evaluator(17, dev)
        run = lambda: ev.counts(pred, gt)
        algo = B * (S * S + 256 * 256 + 3 * 256 * 8)
        what = f"class statistics (accuracy/IoU/Dice inputs) of {B} predictions {S}x{S} vs 256x256 labels per GPU"
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        res = run()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / args.steps     # kernels are launched on torch's current stream
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        out_json = {
            "metric": "images/sec pre-processed" if args.mode == "prep" else "images/sec evaluated",
            "value": round(world * B * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": what, "batch_per_gpu": B, "parallelism": f"batch-split x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(algo / (dev_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(algo / (dev_ms * 1e-3) / 8e12, 4), "traffic": None,
                         "kernel": "resize_h_kernel + resize_v_tensor_kernel" if args.mode == "prep" else "eval_counts_kernel",
                         "device_ms_per_step": round(dev_ms, 4), "algorithmic_bytes_per_step": algo}}
        if not args.no_cpu_baseline:
            if args.mode == "prep":   # what the reference's DataLoader workers run per image: Pillow resize + /255
                from PIL import Image
                n, t1 = 0, time.perf_counter()
                while time.perf_counter() - t1 < 10.0:
                    im = Image.fromarray(host[n % B], "RGB").resize((S, S), Image.BILINEAR)
                    ref = torch.from_numpy(np.asarray(im)).permute(2, 0, 1).contiguous().float().div(255)
                    n += 1
                dt = time.perf_counter() - t1
                ok = bool(torch.equal(ref, res[(n - 1) % B].cpu()))
                out_json["cpu_baseline"] = {"value": round(n / dt, 2), "unit": "images/s", "cores": 1, "kind": "reference",
                                            "sample": f"Pillow {Image.__version__} Image.resize(BILINEAR) + ToTensor on {n} of the "
                                                      f"same photos, one host thread (the reference uses 2 DataLoader workers)"}
                out_json["parity"] = {"bit_exact_vs_pillow": ok}
            else:
                from oracle import preproc_oracle as O
                n, t1 = 0, time.perf_counter()
                p_np, g_np = pred.cpu().numpy(), gt.cpu().numpy()
                while time.perf_counter() - t1 < 10.0:
                    ref = O.image_metrics(p_np[n % B], g_np[n % B], 17)
                    n += 1
                dt = time.perf_counter() - t1
                from visiontransformer_amd.metrics import metrics_from_counts
                got = metrics_from_counts(res[(n - 1) % B].cpu().numpy(), 17, S * S)
                out_json["cpu_baseline"] = {"value": round(n / dt, 2), "unit": "images/s", "cores": 1, "kind": "port",
                                            "sample": f"oracle restatement of datasetTestViTmodel.py:193-219 (numpy) on {n} of the same pairs"}
                out_json["parity"] = {"metrics_identical": bool(all(got[k] == ref[k] for k in ("Accuracy", "Mean_IoU", "Mean_Dice")))}
        print(json.dumps(out_json), flush=True)
    barrier()


# The reference's OWN published workload: nine (P, D, L, A) configurations, batch 4 x 224x224, 17 classes,
# `Inference_Time` = seconds of model(batch) + .sigmoid() per image (model/CE/datasetTestViTmodel.py:97-109,174-186); hardware
# unstated (CPU inferred).  Means of the nine committed model/CE/test/*/*_metrics.csv files.
REF_PUBLISHED_S_PER_IMAGE = {0: 0.3494, 1: 0.1729, 2: 0.6114, 3: 0.4408, 4: 0.8934, 5: 1.4548, 6: 1.4813, 7: 3.1436, 8: 5.8727}


def time_calls(fn, reps, rounds=5):
    """median over `rounds` of the mean wall time of `reps` back-to-back calls (seconds per call)."""
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / reps)
    return float(np.median(out))


def bench_ref_grid(args, rank, world, dev, barrier):
    """--workload ref_grid: the reference's published regime -- each configuration of predict.CONFIGURATIONS at batch 4 (what the
    reference timed) and batch 1 (the worker's single-image call, testViTModel.py:92-126), 224x224, 17 classes: one step = one
    forward of that batch -> fp32 logits + uint8 sigmoid/argmax mask, eager launch sequence and hipGraph replay; s/image beside
    the published figure; FLOPs / time against the MFMA peak of the precision; kernel launches per forward (nodes of the graph);
    the oracle (CPU port of the reference) on the same batch as parity check and CPU baseline."""
    from visiontransformer_amd.config import ViTSegConfig
    from visiontransformer_amd.predict import CONFIGURATIONS
    prec = args.precision
    peak = PEAK_TFLOPS[prec]
    ids = sorted(CONFIGURATIONS) if args.grid_configs is None else [int(v) for v in args.grid_configs.split(",")]
    batches = [4, 1] if args.batch is None else [args.batch]
    grid, cpu_budget = [], 45.0
    for cid in ids:
        P, D, L, A = CONFIGURATIONS[cid]
        cfg = ViTSegConfig(17, P, D, L, A, image_size=224)
        model = ViTSegmentationModel(17, P, D, L, A, image_size=224, device=dev,
                                     precision={"f32": "fp32", "bf16": "bf16", "f16": "fp16", "f32x3": "fp32x3"}[prec]).eval()
        sd_np = synth.make_state_dict(cfg, seed=1)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        entry = {"id": cid, "config": f"P{P}H{D}A{A}", "layers": L, "tokens": cfg.seq_len,
                 "published_s_per_image": REF_PUBLISHED_S_PER_IMAGE[cid], "gflops_per_image": round(cfg.forward_flops_per_image() / 1e9, 2)}
        for B in batches:
            images_np = synth.make_images(cfg, B, seed=0, first_image=rank * B)
            x = torch.from_numpy(images_np).to(dev)
            with torch.no_grad():
                for _ in range(max(args.warmup, 2)):
                    mask, logits = model.predict_mask(x, return_logits=True)
                    model.predict_mask_graphed(x, return_logits=True)
                reps = max(args.steps, 5)
                t_eager = time_calls(lambda: model.predict_mask(x, return_logits=True), reps)
                t_graph = time_calls(lambda: model.predict_mask_graphed(x, return_logits=True), reps)
                gm, gl = model.predict_mask_graphed(x, return_logits=True)
                same = bool(torch.equal(gm, mask) and torch.equal(gl, logits))
            t = min(t_eager, t_graph)
            e = {"ms_eager": round(t_eager * 1e3, 4), "ms_graph": round(t_graph * 1e3, 4), "s_per_image": t / B,
                 "images_per_s": round(B / t, 1), "vs_published": round(REF_PUBLISHED_S_PER_IMAGE[cid] / (t / B), 1),
                 "tflops": round(B * cfg.forward_flops_per_image() / t / 1e12, 2),
                 "frac_of_peak": round(B * cfg.forward_flops_per_image() / t / 1e12 / peak, 4),
                 "launches_per_forward": model.graph_nodes(B, True), "graph_bit_identical": same,
                 "route": model.forward_route(B)}
            if rank == 0 and not args.no_cpu_baseline and B == batches[0] and cpu_budget > 0:
                t0 = time.perf_counter()
                base, parity, _ = cpu_baseline(cfg, sd_np, images_np, logits, mask, seconds_budget=min(cpu_budget, 12.0),
                                               n_images=B)
                cpu_budget -= time.perf_counter() - t0
                e["cpu_baseline"], e["parity"] = base, parity
            entry[f"batch{B}"] = e
        if cid == 0 and not args.no_side_configs:
            # the reference's TRAINING regime (model/CE/trainCurrentViTmodel.py:57: batch 4, 224x224; classes.py:264-297): one
            # step = forward + CE + backward + Adam(lr=1e-5), dropout 0.1 -- informational (these launches are the large-batch
            # kernels; the small-batch route of section 3b is inference-only so far)
            del model
            torch.cuda.empty_cache()
            y4 = torch.from_numpy(synth.make_targets(cfg, 4, seed=0, first_image=0, size=224)).to(dev)
            x4 = torch.from_numpy(synth.make_images(cfg, 4, seed=0, first_image=0)).to(dev)
            entry["train_step_batch4"] = {}
            for tp in ("fp32", "bf16"):
                mt = ViTSegmentationModel(17, P, D, L, A, image_size=224, device=dev, precision=tp, dropout=0.1)
                mt.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
                el, _, loss = run_train_steps(mt, x4, y4, 1, 20, 3, 0, lambda: None)
                entry["train_step_batch4"][tp] = {"ms_per_step": round(el / 20 * 1e3, 3), "images_per_s": round(4 * 20 / el, 1),
                                                  "final_loss": float(loss.detach())}
                del mt
            model = None
        grid.append(entry)
        del model
        torch.cuda.empty_cache()
    barrier()
    if rank == 0:
        head = next((g for g in grid if g["id"] == 0), grid[0])
        hb = head[f"batch{batches[0]}"]
        out = {"metric": f"images/sec (224x224, batch {batches[0]}) ViT seg, the reference's nine published configurations",
               "value": hb["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": min(hb["ms_eager"], hb["ms_graph"]), "higher_is_better": True, "scaling": "weak",
               # BASELINE.md section 1 publishes s/image for exactly these configurations (hardware unstated, CPU inferred)
               "vs_baseline": hb["vs_published"], "dtype": prec, "data": "synthetic",
               "config": {"workload": f"{head['config']} (L{head['layers']}) seg inference, batch {batches[0]} x 224x224, 17 classes, {prec}: "
                                      f"forward -> fp32 logits + uint8 mask (the reference's model/CE/test/*_metrics.csv regime); `grid` "
                                      f"holds all configurations at batch {' and '.join(str(b) for b in batches)}",
                          "batch_per_gpu": batches[0], "image_size": 224, "num_classes": 17},
               "roofline": {"bound": "mfma", "achieved": hb["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": hb["frac_of_peak"],
                            "traffic": None, "kernel": "whole forward (every launch of one step; per-kernel: profiles/r05_ref_grid_*)"},
               "grid": grid}
        if "cpu_baseline" in hb:
            out["cpu_baseline"], out["parity"] = hb["cpu_baseline"], hb["parity"]
        print(json.dumps(out), flush=True)
    barrier()


def run_train_steps(model, x, y, world, steps, warmup, prof_steps, barrier, sync=True):
    """`warmup` untimed + `steps` timed training steps (LightningViTModel.training_step + backward + (N>1: RCCL all-reduce of
    the flat gradient arena) + Adam(lr=1e-5)), then `prof_steps` further steps with hipEvents on the launch stream around
    the GEMM / attention launches (outside the timed region, so the event records do not perturb the timing).
    Returns (seconds of the timed steps on this rank, kernel-group profile, last loss)."""
    from visiontransformer_amd.dist import sync_grads
    from visiontransformer_amd.optim import FusedAdam
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        if sync:
            loss = model.ce_loss(x, y, grad_scale=1.0)   # the factor is folded into the CE gradient: no arena-sized multiply
            loss.backward()
            sync_grads(model)
            opt.step(grad_scale=1.0 / world)
        else:   # the same step with the gradient exchange switched off (model.no_sync(): local gradients only) -- what the
            with model.no_sync():   # all-reduce costs on top of the compute is the difference to the synchronised step
                loss = model.ce_loss(x, y, grad_scale=1.0)
                loss.backward()
            opt.step(grad_scale=1.0)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = {k: v for k, v in _lib.profile_collect().items() if k.startswith("train_")}
    _lib.profile_enable(False)
    return elapsed, prof, loss


def kernel_groups(prof, prof_steps, peak):
    return {k: {"ms_per_step": round(v["ms"] / prof_steps, 3),
                "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else None,
                "frac": round(v["work"] / (v["ms"] * 1e-3) / 1e12 / peak, 4) if v["ms"] > 0 else None}
            for k, v in prof.items()}


def bench_train(args, cfg, model, x, rank, world, dev, barrier):
    """One step = LightningViTModel.training_step + backward + (N>1: RCCL all-reduce of the flat gradient
    arena) + Adam(lr=1e-5): BASELINE configs[2]/[3] (--precision bf16 = mixed precision, --batch 64)."""
    B = args.batch
    y = torch.from_numpy(synth.make_targets(cfg, B, seed=0, first_image=rank * B, size=cfg.image_size)).to(dev)
    PROF_STEPS = 3
    elapsed, prof, loss = run_train_steps(model, x, y, world, args.steps, args.warmup, PROF_STEPS, barrier)
    barrier()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        value = world * B * args.steps / elapsed
        flops_img = 3.0 * cfg.forward_flops_per_image()
        dom = max(prof, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        dom_tflops = d["work"] / (d["ms"] * 1e-3) / 1e12
        names = KERNEL_NAMES.get("train_" + ("f32" if args.precision == "f32" else "bf16"), {})
        peak = PEAK_TFLOPS[args.precision]
        print(json.dumps({
            "metric": "images/sec (512×512) ViT-B/16 seg, 1/2/4/8 MI355X + mask argmax match",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"ViT-B/16 seg TRAINING step (forward + CE + backward + Adam), batch {B}/GPU x "
                                   f"512x512, {args.precision}, dropout {model.dropout} (reference: 0.1)", "batch_per_gpu": B,
                       "global_batch": B * world,
                       "parallelism": f"data-parallel x{world}, bucketed gradient all-reduce ({collective_name(world)}) "
                                      f"overlapped with the backward"},
            # dominant kernel group of the step (largest summed device time): algorithmic FLOPs / hipEvent time on the
            # launch stream; per-kernel times of the same command: profiles/rNN_bench_train_*_kernel_stats.csv
            "roofline": {"bound": "mfma", "achieved": round(dom_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(dom_tflops / peak, 4), "traffic": None, "kernel": names.get(dom, dom),
                         "launches": d["launches"], "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                         "flops_per_launch": d["work"] / max(d["launches"], 1)},
            "kernel_groups": kernel_groups(prof, PROF_STEPS, peak),
            "whole_model": {"flops_per_image": flops_img,
                            "achieved_tflops_per_gpu": round(value / world * flops_img / 1e12, 2),
                            "frac_of_peak": round(value / world * flops_img / 1e12 / peak, 4)},
            "final_loss": float(loss.detach())}), flush=True)
    barrier()


def side_train_bf16(cfg, sd_np, x32, dev, steps=5):
    """BASELINE configs[2] inside the default line (rank 0, N = 1, outside the timed region): ViT-B/16 training step,
    batch 64 x 512x512, bf16 operands / fp32 master weights + Adam, the reference's dropout 0.1."""
    B = 64
    model = ViTSegmentationModel(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                                 cfg.num_attention_heads, image_size=cfg.image_size, precision="bf16", dropout=0.1, device=dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    n0 = x32.shape[0]
    x = torch.cat([x32, torch.from_numpy(synth.make_images(cfg, B - n0, seed=0, first_image=n0)).to(dev)]) if n0 < B else x32[:B]
    y = torch.from_numpy(synth.make_targets(cfg, B, seed=0, first_image=0, size=cfg.image_size)).to(dev)
    PROF_STEPS = 2
    elapsed, prof, loss = run_train_steps(model, x, y, 1, steps, 2, PROF_STEPS, lambda: None)
    peak = PEAK_TFLOPS["bf16"]
    flops_img = 3.0 * cfg.forward_flops_per_image()
    value = B * steps / elapsed
    out = {"workload": "ViT-B/16 seg TRAINING step (forward + CE + backward + Adam), batch 64 x 512x512, bf16 operands / "
                       "fp32 master + Adam, dropout 0.1 (BASELINE.json configs[2])",
           "images_per_s_per_gpu": round(value, 1), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
           "whole_step_frac_of_bf16_peak": round(value * flops_img / 1e12 / peak, 4),
           "kernel_groups": kernel_groups(prof, PROF_STEPS, peak), "final_loss": float(loss.detach())}
    del model, x, y
    torch.cuda.empty_cache()
    return out


def side_train_dist(cfg, sd_np, rank, world, dev, barrier, batch=64, steps=4):
    """N > 1 (every rank, after the timed inference region): what BASELINE configs[3] shards -- the ViT-B/16 training step, batch
    `batch` per rank x 512x512, bf16 operands / fp32 master + Adam, dropout 0.1, data-parallel with the bucketed gradient
    all-reduce (RCCL over xGMI under the driver's launch) overlapped with the backward -- timed with the exchange and, the same
    step, under model.no_sync(); the difference is the communication the overlap does not hide.  Replaces the single-device
    loop of /root/reference/model/CE/createViTmodel.py:68-75."""
    import torch.distributed as dist
    model = ViTSegmentationModel(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                                 cfg.num_attention_heads, image_size=cfg.image_size, precision="bf16", dropout=0.1, device=dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    x = torch.from_numpy(synth.make_images(cfg, batch, seed=0, first_image=rank * batch)).to(dev)
    y = torch.from_numpy(synth.make_targets(cfg, batch, seed=0, first_image=rank * batch, size=cfg.image_size)).to(dev)
    res = {}
    for key, sync in (("ms_per_step", True), ("ms_per_step_no_sync", False)):
        elapsed, _, loss = run_train_steps(model, x, y, world, steps, 2, 0, barrier, sync=sync)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[key] = float(t.item()) / steps * 1e3
    reducer = model._bucket_state()[0] if model._buckets is not None else None
    out = {"workload": f"ViT-B/16 seg TRAINING step (forward + CE + backward + gradient all-reduce + Adam), batch {batch}/GPU x 512x512, "
                       f"bf16 operands / fp32 master + Adam, dropout 0.1 (BASELINE.json configs[3] at {world} ranks)",
           "batch_per_gpu": batch, "global_batch": batch * world, "ranks_seen": dist.get_world_size(),
           "collective": collective_name(world), "backend": str(dist.get_backend()),
           "ms_per_step": round(res["ms_per_step"], 3), "ms_per_step_no_sync": round(res["ms_per_step_no_sync"], 3),
           "exposed_communication_ms": round(res["ms_per_step"] - res["ms_per_step_no_sync"], 3),
           "images_per_s": round(world * batch / (res["ms_per_step"] * 1e-3), 1),
           "gradient_bytes_per_step": int(model.arena.numel()) * 4,
           "all_reduce_messages_per_step": len(reducer.groups) if reducer is not None else 1,
           "grad_sync": model.grad_sync, "final_loss": float(loss.detach())}
    del model, x, y
    torch.cuda.empty_cache()
    return out


def side_ref_grid(dev, reps=10):
    """The reference's own published regime inside the default line (rank 0, N = 1, outside the timed region; the full table with
    the hipGraph column, the oracle check and the CPU baseline per configuration is `--workload ref_grid`): every configuration of
    predict.CONFIGURATIONS at batch 4 x 224x224, 17 classes, fp32 forward -> logits + mask; ViT-B/16 also at batch 1 (the worker's
    call) and its training step at the reference's batch (model/CE/trainCurrentViTmodel.py:57)."""
    from visiontransformer_amd.config import ViTSegConfig
    from visiontransformer_amd.predict import CONFIGURATIONS
    out = {"workload": "fp32 forward (logits + mask) of the nine published configurations, batch 4 x 224x224, 17 classes; "
                       "published = mean Inference_Time per image of model/CE/test/*/*_metrics.csv (hardware unstated)", "configs": {}}
    for cid in sorted(CONFIGURATIONS):
        P, D, L, A = CONFIGURATIONS[cid]
        cfg = ViTSegConfig(17, P, D, L, A, image_size=224)
        model = ViTSegmentationModel(17, P, D, L, A, image_size=224, device=dev).eval()   # random init: a throughput figure
        x = torch.from_numpy(synth.make_images(cfg, 4, seed=0)).to(dev)
        with torch.no_grad():
            for _ in range(3):
                model.predict_mask(x, return_logits=True)
            t4 = time_calls(lambda: model.predict_mask(x, return_logits=True), reps, rounds=3)
            e = {"ms_batch4": round(t4 * 1e3, 3), "frac_of_fp32_peak": round(4 * cfg.forward_flops_per_image() / t4 / 1e12 / PEAK_TFLOPS["f32"], 4),
                 "x_published": round(REF_PUBLISHED_S_PER_IMAGE[cid] / (t4 / 4), 1)}
            if cid == 0:
                for _ in range(3):
                    model.predict_mask(x[:1], return_logits=True)
                e["ms_batch1"] = round(time_calls(lambda: model.predict_mask(x[:1], return_logits=True), reps, rounds=3) * 1e3, 3)
        if cid == 0:
            del model
            # the route's 16-bit form (bf16 operands in the four linears of a block, fp32 everything else)
            mb = ViTSegmentationModel(17, P, D, L, A, image_size=224, precision="bf16", device=dev).eval()
            with torch.no_grad():
                for xb, key in ((x, "ms_batch4_bf16"), (x[:1], "ms_batch1_bf16")):
                    for _ in range(3):
                        mb.predict_mask(xb, return_logits=True)
                    e[key] = round(time_calls(lambda: mb.predict_mask(xb, return_logits=True), reps, rounds=3) * 1e3, 3)
            del mb
            y4 = torch.from_numpy(synth.make_targets(cfg, 4, seed=0, size=224)).to(dev)
            mt = ViTSegmentationModel(17, P, D, L, A, image_size=224, device=dev, dropout=0.1)
            el, _, _ = run_train_steps(mt, x, y4, 1, 10, 3, 0, lambda: None)
            e["train_step_ms_batch4_fp32"] = round(el / 10 * 1e3, 3)
            del mt
            model = None
        out["configs"][f"P{P}H{D}A{A}"] = e
        del model, x
        torch.cuda.empty_cache()
    return out


def side_serve(dev, seconds=6.0):
    """The worker's request path end to end on the device (rows f3 + path + f4 of SURVEY.md section 8 chained; contract:
    /root/reference/backend/core/views.py:97-149, body modelled on model/CE/testViTModel.py:92-126): decoded uint8 photos
    (3024 x 4032 RGB, resident in HBM) -> Preprocessor.images (Pillow-exact Resize((224, 224)) + ToTensor) -> ViT-B/16,
    17 classes, fp32 predict_mask -> Evaluator.counts against a 256 x 256 label map.  Throughput at batch 8, latency per
    single-image request (synchronised after each), and the mask of one request against the oracle run on the same photo
    (Pillow on the host -> oracle forward -> sigmoid / first-max argmax)."""
    from oracle import vitseg_oracle as O
    from visiontransformer_amd.config import ViTSegConfig
    from visiontransformer_amd.metrics import Evaluator
    from visiontransformer_amd.preprocess import Preprocessor
    H, W, S, C, NB = 3024, 4032, 224, 17, 8
    cfg = ViTSegConfig(C, 16, 768, 12, 12, image_size=S)
    sd_np = synth.make_state_dict(cfg, seed=1)
    model = ViTSegmentationModel(C, 16, 768, 12, 12, image_size=S, device=dev).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    rs = np.random.RandomState(11)
    # smooth synthetic "photos" (low-frequency colour fields + noise), so the resize has something to average
    base = rs.randint(0, 256, size=(NB, H // 48, W // 48, 3)).astype(np.uint8)
    host = np.repeat(np.repeat(base, 48, axis=1), 48, axis=2)
    host = np.clip(host.astype(np.int16) + rs.randint(-12, 13, size=host.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    photos = torch.from_numpy(host).to(dev)
    gt = torch.from_numpy(rs.randint(0, C, size=(NB, 256, 256), dtype=np.uint8)).to(dev)
    pre, ev = Preprocessor(S, dev), Evaluator(C, dev)

    def request(lo, hi):
        x = pre.images(photos[lo:hi])
        mask = model.predict_mask(x)
        return mask, ev.counts(mask, gt[lo:hi])

    with torch.no_grad():
        for _ in range(3):
            request(0, NB)
            request(0, 1)
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 2:
            for _ in range(10):
                request(0, NB)
            torch.cuda.synchronize()
            n += 10 * NB
        thr = n / (time.perf_counter() - t0)
        lat = []
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < seconds / 2 or len(lat) < 50:
            i = len(lat) % NB
            ta = time.perf_counter()
            mask1, _ = request(i, i + 1)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - ta)
        mask0, counts0 = request(0, 1)
        torch.cuda.synchronize()
        # the same single-image request with the forward in the small-batch route's 16-bit form (informational: the parity below
        # and every figure above are the fp32 path)
        mb = ViTSegmentationModel(C, 16, 768, 12, 12, image_size=S, precision="bf16", device=dev).eval()
        mb.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        lat16 = []
        for k in range(60):
            ta = time.perf_counter()
            mk16 = mb.predict_mask(pre.images(photos[k % NB:k % NB + 1]))
            ev.counts(mk16, gt[k % NB:k % NB + 1])
            torch.cuda.synchronize()
            if k >= 10:
                lat16.append(time.perf_counter() - ta)
        del mb
    # one request against the oracle: the reference's own host pipeline on the same photo
    from PIL import Image
    im = Image.fromarray(host[0], "RGB").resize((S, S), Image.BILINEAR)
    xr = torch.from_numpy(np.array(im)).permute(2, 0, 1).contiguous().float().div(255)[None]
    with torch.no_grad():
        ref = O.forward(xr, {k: torch.from_numpy(v) for k, v in sd_np.items()}, cfg)
        x_dev = pre.images(photos[0:1])
        _, lg = model.predict_mask(x_dev, return_logits=True)
    err = float((lg.cpu() - ref).abs().max())
    stable = O.mask_stable(ref, 2.0 * err + 1e-7)
    differ = mask0.cpu().long() != O.predict_mask(ref)
    lat_ms = np.sort(np.asarray(lat)) * 1e3
    out = {"workload": f"request path on the device: uint8 photo {H}x{W} (in HBM) -> Resize(({S},{S})) + ToTensor -> ViT-B/16, {C} classes, fp32 "
                       f"forward + sigmoid/argmax mask -> per-image class statistics vs a 256x256 label map",
           "images_per_s_batch8": round(thr, 1), "requests_timed": len(lat),
           "latency_ms_single_image": {"p50": round(float(np.percentile(lat_ms, 50)), 3), "p99": round(float(np.percentile(lat_ms, 99)), 3),
                                       "min": round(float(lat_ms[0]), 3)},
           "latency_ms_single_image_bf16_forward": {"p50": round(float(np.percentile(np.asarray(lat16) * 1e3, 50)), 3)},
           "parity_vs_oracle_one_request": {"input_bit_exact_vs_pillow": bool(torch.equal(x_dev.cpu(), xr)),
                                            "logits_max_abs_err": err, "mask_mismatch_at_stable_pixels": int((differ & stable).sum()),
                                            "unstable_pixels": int((~stable).sum())}}
    del model, photos
    torch.cuda.empty_cache()
    return out


def collective_name(world):
    """What the process group actually runs its all-reduce on (torch's "nccl" backend IS RCCL on ROCm)."""
    if world <= 1:
        return "none: single rank"
    import torch.distributed as dist
    backend = str(dist.get_backend())
    return {"nccl": "RCCL over xGMI"}.get(backend, backend)


def self_launch(n, child=None):
    """`python bench.py --gpus N` without a launcher: N child processes (one rank per GPU, rendezvous on 127.0.0.1),
    started BEFORE this process makes any HIP call (a process that has initialised the GPU must not exec/fork GPU work).
    The parent only waits; rank 0's JSON line goes to stdout unchanged.  Returns the worst child exit code (non-zero as
    soon as any rank dies; the survivors are killed, none is left behind).  `child`: the command line of one rank
    (default: this script with this invocation's arguments; tests/test_dist_cpu.py passes a stub that needs no GPU)."""
    import socket
    import subprocess
    if child is None:
        child = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen(list(child), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):   # a rank died: do not leave the others in a collective
                break
        rc = max(abs(c) for c in (p.poll() for p in procs) if c is not None) if any(p.poll() is not None for p in procs) else 1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                rc = rc or 1
        for p in procs:
            p.wait()          # reap: no zombie / orphan outlives the launcher
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32; 16 for l16_1024_tiled)")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16", "f16", "f32x3"])
    ap.add_argument("--classes", type=int, default=2, help="segmentation classes (BASELINE configs use 2; the reference's "
                                                           "dataset has 17: the decoder tail then writes 17.8 MB/image)")
    ap.add_argument("--grid-configs", default=None, help="ref_grid: comma-separated ids of predict.CONFIGURATIONS (default: all nine)")
    ap.add_argument("--workload", default="b16_512", choices=["b16_512", "l16_1024_tiled", "l16_1024_native", "ref_grid"],
                    help="b16_512 = BASELINE configs[1] (default, the headline metric); l16_1024_tiled = configs[4]; "
                         "l16_1024_native = the same images as one 4097-token sequence each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the configs[2] / configs[4] side runs the default fp32 line appends (train_bf16_path, "
                         "l16_1024_tiled_f16_path)")
    ap.add_argument("--dropout", type=float, default=0.1, help="train mode: dropout probability (reference 0.1)")
    ap.add_argument("--dist-train-batch", type=int, default=64,
                    help="N > 1: images per rank of the data-parallel training step appended to the line (train_bf16_path)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "prep", "eval"],
                    help="train: one step = forward + CE + backward + gradient all-reduce + Adam (fp32)")
    args = ap.parse_args()
    tiled = args.workload in ("l16_1024_tiled", "l16_1024_native")
    if args.precision is None:
        args.precision = "f16" if tiled else "f32"
    if args.batch is None and args.workload != "ref_grid":
        args.batch = 16 if tiled else 32

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU yet (no HIP call so far), so it only
        # starts one fresh child per rank and relays their output; the children take the torch.distributed path below
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VITSEG_LOCAL_DEVICE / VITSEG_DIST_BACKEND: rehearsal of the N > 1 control flow on a one-GPU box (all ranks on the same
    # card over gloo); the driver's multi-GPU runs leave both unset (one rank per GPU, nccl = RCCL)
    local = int(os.environ.get("VITSEG_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (or unset WORLD_SIZE and let "
                         f"bench.py launch them itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VITSEG_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

        world = dist.get_world_size()   # what the process group (RCCL) reports, not what the environment asked for

    def barrier():
        if world > 1:
            dist.barrier()

    if tiled:
        return bench_tiled(args, rank, world, dev, barrier)
    if args.workload == "ref_grid":
        return bench_ref_grid(args, rank, world, dev, barrier)
    if args.mode in ("prep", "eval"):
        return bench_aux(args, rank, world, dev, barrier)

    cfg = vit_base16(num_classes=args.classes, image_size=512)
    B = args.batch
    model = ViTSegmentationModel(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                                 cfg.num_attention_heads, image_size=cfg.image_size,
                                 precision={"f32": "fp32", "bf16": "bf16", "f16": "fp16", "f32x3": "fp32x3"}[args.precision], dropout=args.dropout,
                                 device=dev).eval()
    sd_np = synth.make_state_dict(cfg, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    images_np = synth.make_images(cfg, B, seed=0, first_image=rank * B)  # this rank's shard of the image stream
    x = torch.from_numpy(images_np).to(dev)

    if args.mode == "train":
        return bench_train(args, cfg, model, x, rank, world, dev, barrier)

    with torch.no_grad():
        for _ in range(args.warmup):
            mask, logits = model.predict_mask(x, return_logits=True)
        torch.cuda.synchronize()
        barrier()
        _lib.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mask, logits = model.predict_mask(x, return_logits=True)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)

    split_check = dist_train = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the inference split, seen from every rank: the checksum of each rank's masks (its own shard of the image stream)
        mine = torch.stack([mask.sum(dtype=torch.int64), torch.tensor(mask.numel(), device=dev)])
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        split_check = {"ranks_seen": dist.get_world_size(), "images_per_rank": B,
                       "mask_checksum_per_rank": [int(p[0]) for p in parts], "mask_pixels_per_rank": [int(p[1]) for p in parts]}
        if args.precision == "f32" and not args.no_side_configs:   # the path the north_star shards with a collective
            logits_keep = logits
            dist_train = side_train_dist(cfg, sd_np, rank, world, dev, barrier, batch=args.dist_train_batch)
            logits = logits_keep

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # dominant kernel = the MFMA kernel kind with the largest summed device time in the timed region
        mfma = {k: v for k, v in prof.items() if k.startswith("gemm") or k == "attention"}
        dom = max(mfma, key=lambda k: mfma[k]["ms"])
        d = mfma[dom]
        achieved = d["work"] / (d["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        flops_img = cfg.forward_flops_per_image()
        out = {
            "metric": "images/sec (512×512) ViT-B/16 seg, 1/2/4/8 MI355X + mask argmax match",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"ViT-B/16 seg inference (forward -> fp32 logits + uint8 sigmoid/argmax mask), "
                                   f"batch {B}/GPU x 512x512, {args.precision} (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "image_size": 512, "num_classes": args.classes,
                       "parallelism": f"batch-split x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": pmc_traffic(args.precision, KERNEL_NAMES[args.precision].get(dom, dom), B),
                         "traffic_source": (f"profiles/{os.path.basename(traffic_profile(args.precision))} (committed rocprofv3 "
                                            f"--pmc pass of this command, not measured by this run)"
                                            if traffic_profile(args.precision) and B == 32 else None),
                         "traffic_note": "L2-miss bytes/launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes in "
                                         "profiles/); operand re-reads are served by the 256 MB Infinity Cache",
                         "algorithmic_bytes_per_launch": algorithmic_bytes(args.precision, dom, cfg, B),
                         "kernel": KERNEL_NAMES[args.precision].get(dom, dom), "launches": d["launches"],
                         "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                         "flops_per_launch": d["work"] / max(d["launches"], 1)},
            "whole_model": {"flops_per_image": flops_img,
                            "achieved_tflops_per_gpu": round(value / world * flops_img / 1e12, 2),
                            "frac_of_peak": round(value / world * flops_img / 1e12 / peak, 4)},
            "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if not k.startswith("train_")},
            # the HBM-bound pieces against the 8 TB/s roof (algorithmic bytes / hipEvent time, same timed region)
            "roofline_hbm": hbm_rooflines(prof, ("layernorm", "upsample")),
        }
        if split_check is not None:
            out["inference_split"] = split_check
        if dist_train is not None:
            out["train_bf16_path"] = dist_train
        extras = not args.no_cpu_baseline and world == 1   # CPU baseline / side paths: N = 1 only (spec), rank 0
        if args.precision == "f32" and extras:
            # informational (outside the timed region, rank 0 only): the same step on the other operand formats.
            # f32x3 = fp32 storage, GEMM operands split into half pairs, 3 fp16 MFMAs per product (fp32-grade results);
            # bf16 = bf16 operands / fp32 accumulate.  The timed `value` above is the exact-fp32 MFMA path.
            for prec, key in (("fp32x3", "f32x3_path"), ("bf16", "bf16_path")):
                m2 = ViTSegmentationModel(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                                          cfg.num_attention_heads, image_size=cfg.image_size, precision=prec,
                                          device=dev).eval()
                m2.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
                with torch.no_grad():
                    for _ in range(3):
                        mk2, lg2 = m2.predict_mask(x, return_logits=True)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(10):
                        mk2, lg2 = m2.predict_mask(x, return_logits=True)
                    torch.cuda.synchronize()
                    dt2 = (time.perf_counter() - t1) / 10
                out[key] = {"images_per_s_per_gpu": round(B / dt2, 1), "ms_per_step": round(dt2 * 1e3, 3),
                            "logits_max_abs_diff_vs_f32": float((lg2 - logits).abs().max()),
                            "mask_agreement_vs_f32": float((mk2 == mask).float().mean())}
                if prec == "fp32x3":
                    lg_x3, mk_x3 = lg2, mk2
                del m2
        if extras:
            base, parity, oracle_out = cpu_baseline(cfg, sd_np, images_np, logits, mask)
            out["cpu_baseline"] = base
            out["parity"] = parity
            if args.precision == "f32":   # the split-operand path against the same oracle run
                out["f32x3_path"]["parity_vs_oracle"] = parity_vs(lg_x3, mk_x3, *oracle_out)
                lg_x3 = mk_x3 = lg2 = mk2 = None
        if extras and args.precision == "f32" and B == 32 and args.classes == 2 and not args.no_side_configs:
            # BASELINE configs[2] and configs[4] at one GPU, a few steps each (their own models; the fp32 model and its
            # workspace are released first).  Outside the timed region: `value` is the fp32 inference step above.
            del model
            logits = mask = None
            torch.cuda.empty_cache()
            out["train_bf16_path"] = side_train_bf16(cfg, sd_np, x, dev)
            x = None
            out["l16_1024_tiled_f16_path"] = side_l16_tiled_f16(dev)
            out["serve_path"] = side_serve(dev)
            out["ref_grid_path"] = side_ref_grid(dev)
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: volumes/s of the MST-DINOv2 forward (DinoV2ClassifierSlice) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): a batch of 4 synthetic 64 x 512 x 512 bf16 volumes per GPU.
512 is not a legal input (the reference asserts H % 14 == 0: patch_embed.py:72-73), so each slice is
symmetrically zero-padded to 518 x 518 = 37 x 37 patches, DINOv2's native grid (stated in `config`).
One step = one full forward (patch-embed -> 12 ViT blocks -> slice transformer -> logits) over the
batch, inputs resident in HBM, weights = mst.synth random init of the reference architecture.

N > 1 (weak scaling): the global batch is 4N volumes; every volume's 64 slices are sharded across
the N ranks (each rank encodes 64/N slices of all 4N volumes = the same 256 slices as at N = 1), ONE
RCCL all-gather of the slice embeddings, then the Slice Transformer replicated.

The JSON line also carries `roofline` (dominant kernel, HIP events recorded inside the timed
steps by libmst_hip's profiling hooks) and, at N = 1, `cpu_baseline` (the CPU oracle = PyTorch-CPU
restatement of the reference forward, timed on the host cores over a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "new-vit_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# dense MFMA, MI355X_MICROARCH.md; the fp8 mode issues the NON-scaled v_mfma_f32_16x16x32_fp8_fp8, which runs at the bf16 rate
PEAK = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12, "fp8": 2.5e15}
E, HEADS, DEPTH = 384, 6, 12


def kernel_flops(kind: str, n_slices: int, N: int) -> float:
    """Algorithmic FLOPs of ONE launch over n_slices slices (2 FLOP per MAC)."""
    M = n_slices * N
    return {
        "patch_embed": 2.0 * n_slices * (N - 1) * E * 588,
        "gemm_qkv": 2.0 * M * 3 * E * E,
        "attention": 4.0 * n_slices * N * N * E,
        "gemm_proj": 2.0 * M * E * E,
        "gemm_fc1": 2.0 * M * 4 * E * E,
        "gemm_fc2": 2.0 * M * 4 * E * E,
        "mlp_fused": 4.0 * M * 4 * E * E,
    }.get(kind, 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp8"],
                    help="bf16 is the BASELINE metric's dtype; fp8 = e4m3 linear layers on a bf16 carrier (configs[4])")
    ap.add_argument("--volumes", type=int, default=4, help="volumes per GPU")
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--size", type=int, default=512, help="nominal in-plane size (padded up to a multiple of 14)")
    ap.add_argument("--chunk", type=int, default=0, help="slices per encoder pass (0 = auto)")
    ap.add_argument("--parallelism", default="slice", choices=["slice", "dp"],
                    help="N > 1: 'slice' = slices of every volume sharded over the ranks + one all-gather (north star, default); "
                         "'dp' = whole volumes per rank, no exchange at all (SURVEY 8e: the better choice when B >= N)")
    ap.add_argument("--fp8-calibrate", action="store_true",
                    help="--dtype fp8 only: calibrate static activation scales on the bench batch first (untimed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: using WORLD_SIZE", file=sys.stderr)
    n_gpus = world
    # rehearsal hooks (never set by the driver): all ranks on one GPU over gloo, to exercise the sharded path on a 1-GPU box
    backend = os.environ.get("MST_BENCH_BACKEND", "nccl")
    if os.environ.get("MST_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mst import hip, synth
    from mst.models import DinoV2ClassifierSlice

    hip.load()
    side = (args.size + 13) // 14 * 14          # 512 -> 518
    pad = side - args.size
    D, Bl = args.slices, args.volumes
    B = Bl * n_gpus                              # global batch (weak scaling)
    if D % n_gpus:
        raise SystemExit(f"--slices {D} must be divisible by the number of GPUs {n_gpus}")
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp8": torch.bfloat16}[args.dtype]

    model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=args.dtype,
                                  chunk_slices=args.chunk)
    model.load_state_dict(synth.synth_state_dict("s", 0))
    model = model.to(dev).eval()
    if world > 1 and args.parallelism == "slice":
        model.enable_slice_sharding()

    # synthetic N(0,1) volumes generated on the device (same on every rank), padded 512 -> 518, resident in HBM
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Bdev = Bl if (world > 1 and args.parallelism == "dp") else B      # dp: this rank's own volumes only
    if world > 1 and args.parallelism == "dp":
        g.manual_seed(1 + rank)
    vol = torch.randn((Bdev, 1, D, args.size, args.size), generator=g, device=dev, dtype=torch.float32)
    if pad:
        lo = pad // 2
        vol = torch.nn.functional.pad(vol, (lo, pad - lo, lo, pad - lo))
    vol = vol.to(tdt).contiguous()
    N = 1 + (side // 14) ** 2
    slices_per_rank = B * D // n_gpus

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.fp8_calibrate:
        if args.dtype != "fp8":
            raise SystemExit("--fp8-calibrate needs --dtype fp8")
        model.calibrate_fp8(vol)
    with torch.no_grad():
        for _ in range(args.warmup):
            out = model(vol)
        barrier()
        if not args.no_kernel_timing:
            hip.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(vol)
        barrier()
        dt = time.perf_counter() - t0
        hip.profile_enable(False)
    assert bool(torch.isfinite(out).all()), "non-finite logits"

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    value = B * args.steps / dt

    roofline = None
    kernels = {}
    if not args.no_kernel_timing:
        prof = hip.profile_collect()
        best = None
        for kind, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            launch_slices = slices_per_rank * args.steps * (DEPTH if kind not in ("patch_embed",) else 1) / cnt
            avg_ms = ms / cnt
            fl = kernel_flops(kind, int(round(launch_slices)), N)
            kernels[kind] = {"total_ms": round(ms, 3), "launches": cnt, "avg_ms": round(avg_ms, 4),
                             "tflops": round(fl / (avg_ms * 1e-3) / 1e12, 1) if fl else None}
            if fl and (best is None or ms > best[1]):
                best = (kind, ms, avg_ms, fl)
        if best:
            kind, _, avg_ms, fl = best
            ach = fl / (avg_ms * 1e-3) / 1e12
            roofline = {"kernel": kind, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK[args.dtype] / 1e12,
                        "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK[args.dtype], 4), "traffic": None,
                        "flops_per_launch": fl, "avg_launch_ms": round(avg_ms, 4)}
            pmc = ROOT / "profiles" / "pmc_traffic.json"   # HBM bytes per launch from separate rocprofv3 --pmc passes
            if pmc.exists():
                try:
                    roofline["traffic"] = json.loads(pmc.read_text()).get(args.dtype, {}).get(kind)
                except Exception:
                    pass

    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        from oracle import mst_oracle as O
        sd = synth.synth_state_dict("s", 0)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, int(os.environ.get("MST_CPU_THREADS", 32))))   # beyond ~32 threads MKL/OMP only adds contention here
        torch.set_num_threads(cores)
        with torch.no_grad():
            O.vit_encode(sd, vol[0, 0, :2].float().cpu())         # warm-up
            c0 = time.perf_counter()
            O.vit_encode(sd, vol[0, 0, :2].float().cpu())
            t2 = time.perf_counter() - c0
            ns = int(max(2, min(D, 15.0 / max(t2 / 2, 1e-3))))      # bounded sample: ~15 s of CPU work
            x = vol[0, 0, :ns].float().cpu()
            c0 = time.perf_counter()
            emb, _ = O.vit_encode(sd, x)
            t_enc = time.perf_counter() - c0
            full = emb.repeat((D + ns - 1) // ns, 1)[:D].reshape(1, D, E)
            xs = torch.cat([sd["cls_token"], full], dim=1)
            c0 = time.perf_counter()
            O.slice_fusion(sd, xs)
            t_fus = time.perf_counter() - c0
        per_volume = t_enc / ns * D + t_fus
        cpu_baseline = {"value": round(1.0 / per_volume, 5), "unit": "volumes/s", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": f"oracle/mst_oracle.py (PyTorch-CPU fp32 restatement of the reference forward) on {ns} "
                                  f"of {D} slices at {side}x{side} + the full slice transformer, extrapolated to one volume; "
                                  f"{t_enc:.1f}s encoder + {t_fus:.3f}s fusion"}

    if rank == 0:
        f_vol = 0.0
        try:
            from oracle.mst_oracle import flops_per_volume
            f_vol = flops_per_volume(D, side, side)
        except Exception:
            pass
        line = {
            "metric": "volumes_per_sec_mst_dinov2_fwd", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype + ("-calibrated" if args.fp8_calibrate else ""), "data": "synthetic",
            "config": {"workload": f"MST-DINOv2 (DinoV2ClassifierSlice, ViT-S/14) forward, {Bl} volumes/GPU of "
                                   f"{D}x{args.size}x{args.size} {args.dtype} zero-padded to {side}x{side} (N={N} tokens/slice)",
                       "global_batch_volumes": B, "slices": D, "in_plane": [side, side],
                       "parallelism": "single GPU" if n_gpus == 1 else (
                           f"slice-sharded x{n_gpus} + all-gather of slice embeddings" if args.parallelism == "slice"
                           else f"data-parallel over volumes x{n_gpus}, no exchange"),
                       "weights": "synthetic (mst.synth seed 0), random init of the reference architecture"},
            "achieved_tflops": round(f_vol * value / 1e12, 1) if f_vol else None,
            "mfma_util": round(f_vol * value / (PEAK[args.dtype] * n_gpus), 4) if f_vol else None,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

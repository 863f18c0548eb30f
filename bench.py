#!/usr/bin/env python3
"""Headline benchmark: volumes/s of the MST-DINOv2 forward (DinoV2ClassifierSlice) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment) starts the N ranks itself, as fresh
child processes, BEFORE anything touches a GPU, forwards rank 0's JSON line and exits with the children's status.

Workload (BASELINE.json configs[1]): a batch of 4 synthetic 64 x 512 x 512 bf16 volumes per GPU.
512 is not a legal input (the reference asserts H % 14 == 0: patch_embed.py:72-73), so each slice is
symmetrically zero-padded to 518 x 518 = 37 x 37 patches, DINOv2's native grid (stated in `config`).
One step = one full forward (patch-embed -> 12 ViT blocks -> slice transformer -> logits) over the
batch, inputs resident in HBM, weights = mst.synth random init of the reference architecture.
Volume 0 of rank 0's batch is the input of the reference fixture tests/golden/c3_1x64x518.npz (same weights), so the
logits of the TIMED batch are checked against what the reference itself produced (`parity_check` in the JSON line).

N > 1: `--scaling weak` (default; the contract's definition): the global batch is 4N volumes; every volume's 64 slices
are sharded across the N ranks (each rank encodes 64/N slices of all 4N volumes = the same 256 slices as at N = 1), ONE
RCCL all-gather of the slice embeddings, then the Slice Transformer replicated.  `--scaling strong`
(BASELINE configs[2]: one job, slices sharded): the global batch stays at --volumes and each rank encodes 64/N slices.

The JSON line also carries `roofline` (dominant kernel, HIP events recorded by a caller-owned mst_profiler on every
4th timed step) and, at N = 1, `cpu_baseline` (the CPU oracle = PyTorch-CPU restatement of the reference forward, timed
on the host cores over one whole volume).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "new-vit_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

E, HEADS, DEPTH = 384, 6, 12
# dense MFMA peaks, MI355X_MICROARCH.md.  fp8: libmst_hip issues v_mfma_scale_f32_16x16x128_f8f6f4 (the MX-scaled form, 2x the
# bf16 rate) unless MST_FP8_MX=0 selects the plain 16x16x32_fp8_fp8 (bf16 rate)
PEAK = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12,
        "fp8": 2.5e15 if os.environ.get("MST_FP8_MX") == "0" else 5.0e15}
# the REAL reference (imported in the survey container), fp32, 8 vCPU Xeon @2.6 GHz: 0.25 s per 518^2 slice
SURVEY_REFERENCE_CPU = {"value": 0.06, "unit": "volumes/s", "cores": 8,
                        "note": "reference's own mst/models/dino.py timed in the survey container (SURVEY.md section 6): "
                                "1.01 s per 4 slices at 518x518, i.e. ~16 s per 64-slice volume; not re-measurable on the GPU box "
                                "(the reference does not travel)"}


def flops_per_slice(H: int, W: int, depth: int = DEPTH, e: int = E) -> float:
    """Algorithmic FLOPs of the per-slice encoder, 2 per MAC: 2*Np*E*588 + depth*(24*N*E^2 + 4*N^2*E)."""
    npatch = (H // 14) * (W // 14)
    n = npatch + 1
    return 2.0 * npatch * e * 588 + depth * (24.0 * n * e * e + 4.0 * n * n * e)


def flops_per_volume(D: int, H: int, W: int, e: int = E) -> float:
    """F_vol = D * F_slice + F_fusion, F_fusion = 2L*3E^2 + 4L^2*E + 2L*E^2 + 4L*E^2 + 4E, L = D + 1."""
    L = D + 1
    return D * flops_per_slice(H, W, e=e) + 2.0 * L * 3 * e * e + 4.0 * L * L * e + 2.0 * L * e * e + 4.0 * L * e * e + 4.0 * e


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
        "mlp_fused": 4.0 * M * 4 * E * E,               # fc1 + fc2 (+ the folded out-projection: see block_fused)
        "block_fused": 4.0 * M * 4 * E * E + 2.0 * M * E * E,   # out-projection + fc1 + fc2 in one launch
    }.get(kind, 0.0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp8"],
                    help="bf16 is the BASELINE metric's dtype; fp8 = e4m3 linear layers on a bf16 carrier (configs[4])")
    ap.add_argument("--volumes", type=int, default=4, help="volumes per GPU (weak) / in total (strong)")
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--size", type=int, default=512, help="nominal in-plane size (padded up to a multiple of 14)")
    ap.add_argument("--chunk", type=int, default=0, help="slices per encoder pass (0 = auto)")
    ap.add_argument("--parallelism", default="slice", choices=["slice", "dp"],
                    help="N > 1: 'slice' = slices of every volume sharded over the ranks + one all-gather (north star, default); "
                         "'dp' = whole volumes per rank, no exchange at all (SURVEY 8e: the better choice when B >= N)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --volumes per GPU (global batch grows with N); strong: --volumes in total, 64/N slices per rank")
    ap.add_argument("--fp8-calibrate", action="store_true",
                    help="--dtype fp8 only: calibrate static activation scales on the bench batch first (untimed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--prune-last-block", action="store_true",
                    help="NOT the headline configuration: the last block computes only what is read behind it (K/V of every "
                         "token, attention + MLP of the class tokens).  Same results; FLOPs are then counted as executed")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """--gpus N without a launcher: start N ranks as fresh children (one device each) before any GPU call here."""
    import torch            # device_count() does not initialise the GPU on this image
    single = os.environ.get("MST_BENCH_SINGLE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not single:
        print(f"[bench] --gpus {args.gpus} requested but only {have} device(s) are visible", file=sys.stderr)
        return 3
    env = dict(os.environ)
    # dmabuf is the only IPC mode the host driver of this pool supports: without it RCCL's buffer exchange between the ranks fails with
    # "hipIpcGetMemHandle: invalid argument" (the image exports it already; kept explicit for launches from a clean environment)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    port = env.get("MASTER_PORT") or str(29400 + os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode      # rank 0's JSON line goes straight to our stdout


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE {world} ranks", file=sys.stderr)
        sys.exit(2)
    n_gpus = world
    # rehearsal hooks (never set by the driver): all ranks on one GPU over gloo, to exercise the sharded path on a 1-GPU box
    backend = os.environ.get("MST_BENCH_BACKEND", "nccl")
    single = os.environ.get("MST_BENCH_SINGLE_DEVICE") == "1"
    if single:
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
    B = Bl * n_gpus if args.scaling == "weak" else Bl          # global batch
    slice_par = world > 1 and args.parallelism == "slice"
    if slice_par and D % n_gpus:
        raise SystemExit(f"--slices {D} must be divisible by the number of GPUs {n_gpus}")
    if world > 1 and args.parallelism == "dp" and B % n_gpus:
        raise SystemExit(f"data-parallel: global batch {B} must be divisible by {n_gpus}")
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp8": torch.bfloat16}[args.dtype]

    # the reference fixture rides in the timed batch when the workload has its shape (64 x 518^2, default weights)
    fixture = None
    gpath = ROOT / "tests" / "golden" / "c3_1x64x518.npz"
    if not args.no_parity_check and gpath.exists() and (D, side) == (64, 518):
        import numpy as np
        with np.load(gpath) as z:
            fixture = {"seed": int(z["seed"]), "logits": torch.from_numpy(z["logits"].copy())}
    wseed = fixture["seed"] if fixture else 0

    model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype=args.dtype,
                                  chunk_slices=args.chunk, prune_last_block=args.prune_last_block)
    model.load_state_dict(synth.synth_state_dict("s", wseed))
    model = model.to(dev).eval()
    if slice_par:
        if backend == "nccl":
            model.enable_slice_sharding()
        else:
            sys.path.insert(0, str(ROOT / "tools"))
            from rehearsal import HostStagedSharding
            model.enable_slice_sharding(sharding=HostStagedSharding())

    # synthetic N(0,1) volumes generated on the device (same on every rank), padded 512 -> 518, resident in HBM
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    dp = world > 1 and args.parallelism == "dp"
    Bdev = B // n_gpus if dp else B                            # dp: this rank's own volumes only
    if dp:
        g.manual_seed(1 + rank)
    vol = torch.randn((Bdev, 1, D, args.size, args.size), generator=g, device=dev, dtype=torch.float32)
    if pad:
        lo = pad // 2
        vol = torch.nn.functional.pad(vol, (lo, pad - lo, lo, pad - lo))
    if fixture and (not dp or rank == 0):
        vol[0] = synth.synth_volume((1, 1, 64, 518, 518), fixture["seed"] + 100)[0].to(dev)
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
    prof = None if args.no_kernel_timing else hip.Profiler()
    sampled_steps = 0
    with torch.no_grad():
        for _ in range(args.warmup):
            out = model(vol)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # kernel events on every 4th timed step only: the brackets cost ~0.4 ms per fully bracketed step
            sample = prof is not None and (i % 4 == 0)
            model.profiler = prof if sample else None
            sampled_steps += int(sample)
            out = model(vol)
        barrier()
        dt = time.perf_counter() - t0
    model.profiler = None
    assert bool(torch.isfinite(out).all()), "non-finite logits"

    parity = None
    if fixture and rank == 0:
        tol = {"bf16": 3e-2, "fp16": 5e-3, "fp32": 1e-4, "fp8": 2.5e-1}[args.dtype]   # tests/test_model_gpu.py TOL
        err = float((out[0].float().cpu() - fixture["logits"][0]).abs().max())
        parity = {"fixture": "tests/golden/c3_1x64x518.npz (reference output)", "what": "logits of volume 0 of the timed batch",
                  "max_abs_err": round(err, 6), "tol": tol, "ok": bool(err < tol)}
        assert parity["ok"], f"timed batch deviates from the reference fixture: {parity}"

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            tc = t.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            t = tc
    dt = float(t.item())
    value = B * args.steps / dt

    roofline = None
    kernels = {}
    if prof is not None:
        best = None
        for kind, (ms, cnt) in prof.collect().items():
            if cnt == 0:
                continue
            layers = 1 if kind == "patch_embed" else DEPTH
            if args.prune_last_block and kind == "block_fused":
                layers = DEPTH - 1                       # the last block runs on the class-token rows only
            launch_slices = slices_per_rank * sampled_steps * layers / cnt
            avg_ms = ms / cnt
            kname = kind
            fl = kernel_flops(kname, int(round(launch_slices)), N)
            if args.prune_last_block and kind in ("attention", "gemm_proj", "gemm_fc1", "gemm_fc2", "layernorm"):
                fl = 0.0                                 # these kinds mix (or are only) class-token launches: no per-launch rate
            kernels[kname] = {"total_ms": round(ms, 3), "launches": cnt, "avg_ms": round(avg_ms, 4),
                              "tflops": round(fl / (avg_ms * 1e-3) / 1e12, 1) if fl else None}
            if fl and (best is None or ms > best[1]):
                best = (kname, ms, avg_ms, fl)
        prof.close()
        if best:
            kind, _, avg_ms, fl = best
            ach = fl / (avg_ms * 1e-3) / 1e12
            roofline = {"kernel": kind, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK[args.dtype] / 1e12,
                        "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK[args.dtype], 4), "traffic": None,
                        "flops_per_launch": fl, "avg_launch_ms": round(avg_ms, 4),
                        "sampled_steps": sampled_steps}
            # HBM bytes per launch from separate rocprofv3 --pmc passes: only for the exact workload they were taken on
            pmc = ROOT / "profiles" / "pmc_traffic.json"
            if pmc.exists():
                try:
                    for ent in json.loads(pmc.read_text()).get("entries", []):
                        if (ent["dtype"], ent["kernel"], ent["slices_per_launch"], ent["side"]) == \
                                (args.dtype, kind, slices_per_rank, side):
                            roofline["traffic"] = ent["bytes_per_launch"]
                            roofline["traffic_source"] = ent.get("source")
                except Exception:
                    pass
            # matrix-pipe busy fraction of the same kernel from a separate rocprofv3 SQ pass (SQ_VALU_MFMA_BUSY_CYCLES / (kernel
            # cycles x SIMDs)): the counter view of `frac`, quoted only for the exact workload it was taken on
            sqf = ROOT / "profiles" / "sq_counters.json"
            if sqf.exists():
                try:
                    for ent in json.loads(sqf.read_text()).get("entries", []):
                        if (ent["dtype"], ent["kernel"], ent["slices_per_launch"], ent["side"]) == \
                                (args.dtype, kind, slices_per_rank, side):
                            roofline["mfma_busy_frac"] = ent["mfma_busy_frac"]
                            roofline["wave_time_split"] = ent.get("wave_time_split")
                            roofline["counters_source"] = ent.get("source")
                except Exception:
                    pass

    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        from oracle import mst_oracle as O          # the checker, timed as the CPU baseline (kind: port)
        sd = synth.synth_state_dict("s", wseed)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, int(os.environ.get("MST_CPU_THREADS", 32))))   # beyond ~32 threads MKL/OMP only adds contention here
        torch.set_num_threads(cores)
        with torch.no_grad():
            O.vit_encode(sd, vol[0, 0, :2].float().cpu())         # warm-up
            x = vol[0, 0].float().cpu()                            # ALL D slices of one volume
            c0 = time.perf_counter()
            parts = [O.vit_encode(sd, x[i:i + 8])[0] for i in range(0, D, 8)]
            t_enc = time.perf_counter() - c0
            xs = torch.cat([sd["cls_token"], torch.cat(parts).reshape(1, D, E)], dim=1)
            c0 = time.perf_counter()
            O.slice_fusion(sd, xs)
            t_fus = time.perf_counter() - c0
        cpu_baseline = {"value": round(1.0 / (t_enc + t_fus), 5), "unit": "volumes/s", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": f"oracle/mst_oracle.py (PyTorch-CPU fp32 restatement of the reference forward) on one whole "
                                  f"volume: all {D} slices at {side}x{side} + the slice transformer; "
                                  f"{t_enc:.1f}s encoder + {t_fus:.3f}s fusion",
                        "reference_survey": SURVEY_REFERENCE_CPU}

    if rank == 0:
        f_vol = flops_per_volume(D, side, side)
        if args.prune_last_block:                        # executed, not algorithmic: the last block keeps its K/V projection only
            f_vol -= D * (4.0 * N * N * E + 18.0 * N * E * E)      # attention + out-projection + MLP of the patch tokens
        par = "single GPU"
        if n_gpus > 1:
            par = (f"slice-sharded x{n_gpus} ({D // n_gpus} slices of every volume per rank) + all-gather of slice embeddings"
                   if args.parallelism == "slice" else f"data-parallel over volumes x{n_gpus}, no exchange")
        line = {
            "metric": "volumes_per_sec_mst_dinov2_fwd", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype + ("-calibrated" if args.fp8_calibrate else ""), "data": "synthetic",
            "config": {"workload": f"MST-DINOv2 (DinoV2ClassifierSlice, ViT-S/14) forward, "
                                   f"{Bl} volumes{'/GPU' if args.scaling == 'weak' else ' in total'} of "
                                   f"{D}x{args.size}x{args.size} {args.dtype} zero-padded to {side}x{side} (N={N} tokens/slice)",
                       "global_batch_volumes": B, "slices": D, "in_plane": [side, side], "parallelism": par,
                       "weights": f"synthetic (mst.synth seed {wseed}), random init of the reference architecture",
                       **({"prune_last_block": "opt-in: dead patch-token work of the last block skipped (same outputs); FLOPs counted "
                                               "as executed; not the headline configuration"} if args.prune_last_block else {})},
            "achieved_tflops": round(f_vol * value / 1e12, 1),
            "mfma_util": round(f_vol * value / (PEAK[args.dtype] * n_gpus), 4),
            "parity_check": parity,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

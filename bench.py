#!/usr/bin/env python
"""
bench.py — throughput of the SVI training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

Workload (BASELINE.json configs[1]): iVAE, data_dim (28, 28), latent_dim 2, invariances ['r','t'],
Bernoulli likelihood, batch 256 PER GPU (weak scaling), bf16 as that config names it: the decoder's hidden-layer
contractions take bf16 operands on the MFMA with fp32 accumulation (--fused 3: ELBO within 1e-5 of the fp32 oracle,
gradients to ~1e-2 — mixed-precision training), everything else fp32.  The same run then repeats the measurement on
the library's default fp32-class path (--fused 2, "bf16x3": hi+lo operands, 3 products — the 1e-4 parity mode for
gradients too) and reports it as `fp32_class`; --fused 1 selects the f32-input MFMA kernel, --fused 0 the
layer-by-layer path.  Synthetic data torch.rand(..., seed 0), model/trainer seed 1, random-init weights.
A step = Trace_ELBO loss + gradients over one minibatch already resident in HBM
(pv_ivae_loss_and_grads) + [one all-reduce of the flat gradient when N > 1] + Adam (pv_adam_step).

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     the dominant kernel (the fused decoder kernel) against the dense MFMA peak of the instruction it
               runs on, its duration measured with HIP events recorded on the launch stream inside the timed
               region (every 8th step: an event pair costs ~10 us of stream bubbles),
  cpu_baseline the CPU oracle (eager-torch restatement of the reference) timed on this host's cores
               on a bounded sample of the same workload (rank 0, N = 1 only),
  elbo         per-image loss of the first timed step next to the oracle's value on the same inputs.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DATA_DIM = (28, 28)
INVARIANCES = ["r", "t"]
LATENT_DIM = 2
BATCH_PER_GPU = 256
N_RING = 16                      # distinct resident minibatches (n = 16 * B)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* f32-in peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0
# HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE doubled per the guide's
# gfx950 correction + WRITE_SIZE, KiB -> bytes), batch 256: see profiles/r01*_pmc_*.txt.  None: not collected.
TRAFFIC_BYTES = {1: 2 * 6982 * 1024 + 50298 * 1024, 2: 2 * 1331 * 1024 + 37943 * 1024,
                 3: 2 * 994 * 1024 + 37925 * 1024}   # 2, 3: profiles/r01i_pmc_* (same values in r01h, r01j)


def decoder_flops_per_image(n_pix, hidden=128, coord_dim=2):
    """Algorithmic fwd+bwd FLOPs of the spatial decoder per image:
    per pixel fwd = 2*(cd*H) + 2*H*H + 2*H*H + 2*H ; fwd+bwd = 3x."""
    per_pix = 2 * coord_dim * hidden + 2 * hidden * hidden * 2 + 2 * hidden
    return 3 * n_pix * per_pix


def encoder_flops_per_image(n_pix, z_dim, hidden=128, latent=2):
    return 3 * (2 * (n_pix * hidden + hidden * hidden + 2 * hidden * z_dim) + 2 * hidden * latent)


EV_EVERY = 8


class HipEvents:
    """Raw hipEvent_t pairs (the library records them on the stream it launches on)."""
    def __init__(self, n):
        self.hip = C.CDLL("libamdhip64.so.7")      # already loaded by torch: the same runtime instance
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.pairs = []
        for _ in range(n):
            a, b = C.c_void_p(), C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(a)) == 0 and self.hip.hipEventCreate(C.byref(b)) == 0
            self.pairs.append((a, b))

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = C.c_float()
            if self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0:
                out.append(ms.value)
        return out


def cpu_baseline(budget_s=15.0):
    """The CPU oracle (a restatement of the reference's eager-torch path) on the same workload."""
    import pyroved_amd as pv
    from oracle import svi_oracle as orc
    ncores = os.cpu_count() or 1
    model = pv.models.iVAE(DATA_DIM, LATENT_DIM, INVARIANCES, seed=1, device="cpu")
    cfg = orc.Config(data_dim=DATA_DIM, latent_dim=LATENT_DIM, invariances=INVARIANCES)
    o = orc.SVIOracle(model.state_dict(), cfg)
    x = torch.rand(BATCH_PER_GPU, *DATA_DIM, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(1)
    eps0 = torch.empty(BATCH_PER_GPU, cfg.z_dim).normal_()
    torch.set_num_threads(min(16, ncores))
    loss0 = o.step(x, eps0)                    # also the warm-up step
    # eager torch does not scale to every hardware thread of a big host on (200704 x 128) operands:
    # probe a few thread counts (one step each) and time the best one
    best_t, best_n = None, None
    for nt in (8, 16, 32, 64, 128):
        if nt > ncores:
            break
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        o.step(x, o.draw_eps(BATCH_PER_GPU))
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if dt > 2.5 * best_t:
            break
    torch.set_num_threads(best_n)
    t0 = time.perf_counter()
    n = 0
    while True:
        o.step(x, o.draw_eps(BATCH_PER_GPU))
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 3) or n >= 200:
            break
    return dict(value=n * BATCH_PER_GPU / el, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d SVI steps of batch %d (%.1f s) of the same iVAE 28x28 ['r','t'] workload, eager torch CPU "
                       "oracle (oracle/svi_oracle.py), %d threads" % (n, BATCH_PER_GPU, el, torch.get_num_threads()),
                ms_per_step=1e3 * el / n), loss0 / BATCH_PER_GPU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--fused", type=int, default=int(os.environ.get("PV_BENCH_FUSED", "3")),
                    help="0 layered, 1 fused f32 MFMA, 2 fused bf16x3 (fp32-class), 3 fused plain bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the second leg (the fp32-class path) of the default run")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--two-call", action="store_true", help="N=1: pv_ivae_loss_and_grads + pv_adam_step instead of pv_ivae_step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # test hook (one-GPU boxes): PV_BENCH_BACKEND=gloo PV_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 over gloo, to
    # exercise the multi-rank code path end to end where no second GPU exists; the numbers then mean nothing
    backend = os.environ.get("PV_BENCH_BACKEND", "nccl")
    if os.environ.get("PV_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as td
    if world > 1:
        if backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group(backend)

    import pyroved_amd as pv
    from pyroved_amd import dist as pvdist

    B = args.batch
    n_pix = DATA_DIM[0] * DATA_DIM[1]

    def run(fused):
        """warm-up + the timed region on one decoder path; -> (elapsed s (max over ranks), kernel ms samples, losses, engine)"""
        return _run(args, fused, pv, pvdist, td, dev, rank, world, B, n_pix)

    elapsed, kms, losses, eng, model = run(args.fused)
    alt = None
    if args.fused == 3 and not args.no_alt:
        # the same workload on the fp32-class path (bf16 split precision), reported next to the headline
        a_el, a_kms, a_losses, _, _ = run(2)
        alt = {"path": "fused-bf16x3", "dtype": "bf16x3", "value": args.steps * B * world / a_el, "unit": "images/s",
               "ms_per_step": 1e3 * a_el / args.steps, "kernel_ms": sum(a_kms) / max(len(a_kms), 1),
               "loss_per_image_step0": a_losses[0].item() / (B * world)}
    _report(args, rank, world, B, n_pix, elapsed, kms, losses, eng, model, alt)
    if world > 1:
        td.destroy_process_group()


def _run(args, fused, pv, pvdist, td, dev, rank, world, B, n_pix):
    model = pv.models.iVAE(DATA_DIM, LATENT_DIM, INVARIANCES, seed=1, device=dev)
    eng = model.engine(fused=fused)
    if world > 1:
        pvdist.broadcast_(eng.flat)
    # synthetic data, resident in HBM before the timed region; every rank gets its own shard of a
    # global ring of N_RING * world minibatches (weak scaling: B per GPU)
    g = torch.Generator().manual_seed(0)
    ring = torch.rand(N_RING * world * B, *DATA_DIM, generator=g)
    ring = ring.view(N_RING, world, B, n_pix)[:, rank].contiguous().to(dev)
    total_steps = args.warmup + args.steps
    torch.manual_seed(1)
    eps_all = torch.empty(total_steps, world, B, model.z_dim).normal_()[:, rank].contiguous().to(dev)
    # the dominant kernel is bracketed with HIP events on every EV_EVERY-th timed step (an event pair costs ~10 us
    # of stream bubbles around the kernel it brackets: sampled so that the clock measures the path, not the probe)
    n_ev = (args.steps + EV_EVERY - 1) // EV_EVERY
    events = HipEvents(n_ev)
    hist = torch.zeros(total_steps, 4, device=dev)

    def step(i, timed_idx=None):
        sampled = timed_idx is not None and timed_idx % EV_EVERY == 0
        eng.events = events.pairs[timed_idx // EV_EVERY] if sampled else (None, None)
        if world > 1:
            eng.loss_and_grads(ring[i % N_RING], eps_all[i])
            pvdist.allreduce_sum_(eng.grad)           # gradients + the 4 loss scalars in one collective
            hist[i].copy_(eng.scalars)
            eng.adam_step()
        elif args.two_call:
            eng.loss_and_grads(ring[i % N_RING], eps_all[i], scalars_out=hist[i])   # loss lands in the history
            eng.adam_step()
        else:
            # single process: SVI.step as ONE library call (pv_ivae_step: ELBO + gradients + Adam; the update rides in
            # the last gradient launch — bit-identical to the two calls, tests/test_gpu_parity.py)
            eng.loss_and_grads(ring[i % N_RING], eps_all[i], scalars_out=hist[i], step=True)

    # one untimed, state-free pass first (gradients only, no optimizer update, no collective on real data): module
    # loading and first-launch costs never land in the timed region, whatever --warmup is
    eng.loss_and_grads(ring[0], eps_all[0])
    if world > 1:
        pvdist.allreduce_sum_(torch.zeros_like(eng.grad))
    for i in range(args.warmup):
        step(i)
    if world > 1:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, i)
    if world > 1:
        td.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    elapsed = t.item()

    return elapsed, events.elapsed_ms(), hist[:, 0].cpu(), eng, model


def _report(args, rank, world, B, n_pix, elapsed, kms, losses, eng, model, alt):
    if rank == 0:
        images = args.steps * B * world
        value = images / elapsed
        dec_fl = decoder_flops_per_image(n_pix)
        k_avg_ms = sum(kms) / max(len(kms), 1)
        # which kernel the events bracket depends on the path (see pv_plan.hip / pv_sdec_fused.hip)
        fused_used = eng.uses_fused(B)
        mode = args.fused if fused_used else 0
        if mode == 2:
            # split-precision bf16 MFMA: `achieved` counts the ALGORITHMIC fp32 FLOPs once, although
            # the kernel issues 3 bf16 MFMAs per product; the peak is the dense bf16 MFMA peak it runs on
            kname = "pv_sdec_fused_bf16_kernel (decoder fwd+bwd, all layers, bf16x3 split precision)"
            flops_per_launch, peak, dtype = dec_fl * B, MFMA_BF16_PEAK_TFLOPS, "bf16x3"
            arith = "bf16 split-precision MFMA (hi+lo, 3 products, fp32 accumulate; fp32 elsewhere)"
        elif mode == 3:
            kname = "pv_sdec_fused_bf16_kernel<X3=false> (decoder fwd+bwd, all layers, plain bf16 operands)"
            flops_per_launch, peak, dtype = dec_fl * B, MFMA_BF16_PEAK_TFLOPS, "bf16"
            arith = "bf16 MFMA operands for the hidden-layer contractions (fp32 accumulate; fp32 elsewhere)"
        elif mode == 1:
            kname = "pv_sdec_fused_kernel (decoder fwd+bwd, all layers, f32-input MFMA)"
            flops_per_launch, peak, dtype = dec_fl * B, MFMA_F32_PEAK_TFLOPS, "f32"
            arith = "fp32 (f32-input MFMA)"
        else:
            kname = "pv_gemm_kernel<NT> (decoder hidden layer fwd, M=B*N, K=N=128)"
            flops_per_launch, peak, dtype = 2.0 * B * n_pix * 128 * 128, MFMA_F32_PEAK_TFLOPS, "f32"
            arith = "fp32 (f32-input MFMA)"
        achieved = flops_per_launch / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0
        out = {
            "metric": "images/sec (SVI step)", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "iVAE 28x28 invariances=['r','t'] latent_dim=2 bernoulli, batch %d per GPU "
                                   "(global %d), %s, SVI step = ELBO+grads+%sAdam"
                                   % (B, B * world, arith, "allreduce+" if world > 1 else ""),
                       "parallelism": "dp%d" % world, "path": {0: "layered", 1: "fused-f32", 2: "fused-bf16x3", 3: "fused-bf16"}[mode]},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": TRAFFIC_BYTES.get(mode), "kernel": kname,
                         "kernel_ms": k_avg_ms, "flops_per_launch": flops_per_launch,
                         "frac_of_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS},
            "step_frac_of_f32_mfma_roof": value / world * (dec_fl + encoder_flops_per_image(n_pix, model.z_dim))
            / (MFMA_F32_PEAK_TFLOPS * 1e12),
            "elbo": {"loss_per_image_first_timed_step": losses[args.warmup].item() / (B * world),
                     "loss_per_image_last_step": losses[-1].item() / (B * world),
                     "loss_per_image_step0": losses[0].item() / (B * world)},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, loss0 = cpu_baseline()
            out["cpu_baseline"] = cb
            out["elbo"]["oracle_loss_per_image_step0"] = loss0
            if B == BATCH_PER_GPU:
                out["elbo"]["rel_err_step0"] = abs(out["elbo"]["loss_per_image_step0"] - loss0) / abs(loss0)
        if alt is not None:
            out["fp32_class"] = alt
        print(json.dumps(out))


if __name__ == "__main__":
    main()

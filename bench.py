#!/usr/bin/env python3
"""Headline benchmark: images/sec of the denoise forward at batch 256, 128x128x3, fp32, per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1.  Started as a plain command (no WORLD_SIZE in the environment) with --gpus N > 1, this process
makes no GPU call: it starts N children of itself — one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
a free MASTER_PORT set — relays rank 0's single JSON line to its own stdout and exits non-zero if any child does
(`launch_ranks`).  Under torch.distributed.run the ranks already exist and each runs the benchmark directly.

A "step" is one forward of the hot path over one batch of 256 synthetic 128x128x3 Gaussian-noised
images per GPU (BASELINE.json configs[1]; with N GPUs the global batch is 256*N — configs[2] at
N=8 — sharded contiguously, weak scaling), inputs already resident in HBM.  Rank 0 loads the
(seeded synthetic) weights and ONE RCCL broadcast of the packed blob distributes them; the
forward itself has no collective.  Prints one JSON line on rank 0.

Extra objects on that line:
  roofline      dominant kernel of the forward: FLOPs the matrix pipe EXECUTES per launch (Winograd launches: 24/72 or 16/36 of
                the direct-convolution count) / its mean launch duration from HIP events recorded on the launch stream
                during the timed steps; peak = gfx950 dense fp32 matrix rate (157.3 TFLOP/s); `frac` <= 1 by
                construction; `achieved_algorithmic` = direct-convolution FLOPs / the same time
  layers        every launch: ms, executed and algorithmic TFLOP/s, GB/s, fraction of its own (mfma|hbm) roofline
  configs       (N=1 only) short legs at BASELINE.json configs[3] (B=256, 256x256, fp32) and configs[4] (B=512, fp16
                storage): images/sec, slowest launch, parity spot check
  hot_weights_leg, dist_world1_leg  (N=1 only) 5 steps on the He-gain weight set; RCCL communicator + weights broadcast at world size 1
  split16_leg   (N=1 only) the opt-in split-operand algorithm of the 3x3 layers (conv_algo="split16") on the same batch: images/sec and error on both weight sets
  rccl, parity.max_abs_err_all_ranks  (N>1, or --force-dist) what RCCL reports for the weights communicator and the one broadcast; the worst
                error over ALL ranks of two images of each rank's own shard against the CPU oracle (ranks > 0 run on the broadcast weights)
  rehearsal     only with --rehearse-shared-gpu: the N>1 code path with all ranks on cuda:0 over gloo (one-GPU boxes); not a scaling number
  cpu_baseline  (N=1 only) the CPU oracle = the reference's forward re-stated on the ATen CPU
                operators the reference itself calls, timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def launch_ranks(n, argv, child_cmd=None, poll_s=0.2, out=None):
    """`python bench.py --gpus N` as a plain command: be the launcher of N ranks, one per GPU of this node.

    The calling process has made no GPU call (nothing GPU-related is even imported yet) and never becomes a rank or replaces
    itself: it starts N fresh children (`child_cmd` + `argv`; default this interpreter on this file), each with RANK = LOCAL_RANK = r,
    WORLD_SIZE = LOCAL_WORLD_SIZE = n, MASTER_ADDR = 127.0.0.1 and one free MASTER_PORT, relays what rank 0 writes to its stdout
    (the contract's ONE JSON line) to `out` (default this process's stdout), sends the other ranks' stdout to stderr, and
    returns 0 only if every child exited 0.  When one child fails the others are terminated (exact PIDs) — they would wait
    for it in a collective — and the first non-zero exit code is returned."""
    out = out or sys.stdout
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = list(child_cmd) if child_cmd is not None else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else 2))   # 2 = this process's stderr
    rc = 0
    try:
        import threading

        def relay():   # rank 0 prints its one line at the very end; read in a thread so a failing peer is still noticed
            for line in procs[0].stdout:
                out.write(line.decode(errors="replace"))
                out.flush()

        t = threading.Thread(target=relay, daemon=True)
        t.start()
        live = set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 128 - code
                        print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        break
            if live and rc == 0:
                time.sleep(poll_s)
        if rc == 0:
            t.join(timeout=30)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def _import_runtime():
    """numpy / torch / the package, imported only by a process that IS a rank (the launcher above stays GPU-free)."""
    global np, torch, cid, cdist, synth, launch_table
    import numpy as np
    import torch

    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import dist as cdist
    from celebrity_image_denoiser_amd import synth
    from celebrity_image_denoiser_amd.generator import launch_table


PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0          # spec; ~6300 GB/s achievable
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16 matrix rate (same guide)


def host_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sd, budget_s: float = 15.0):
    """Time the CPU oracle (kind "port": the reference forward re-stated on ATen CPU ops) on a bounded
    sample of the same workload: batches of 32 images 128x128 until ~budget_s of CPU work."""
    from oracle import torch_oracle

    cores = host_cores()
    torch.set_num_threads(cores)
    x, _, _ = synth.make_batch(32, 128, 128, first_index=5000)
    torch_oracle.forward(sd, x[:4])   # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    torch_oracle.forward(sd, x)
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(16, budget_s / max(t1, 1e-3) - 1)))
    t0 = time.perf_counter()
    for _ in range(reps):
        torch_oracle.forward(sd, x)
    dt = time.perf_counter() - t0
    out = {"value": round(32 * reps / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{reps} x batch of 32 images 128x128x3 fp32 through oracle/torch_oracle.py (ATen CPU conv2d/conv_transpose2d/max_pool2d, the operators the reference module calls), {dt:.1f} s"}
    try:   # second opinion: the dependency-free C restatement (OpenMP), 8 images
        from oracle import c_oracle

        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        t0 = time.perf_counter()
        c_oracle.forward(sd, x[:8])
        out["c_oracle_images_per_sec"] = round(8 / (time.perf_counter() - t0), 2)
    except Exception as e:  # pragma: no cover
        out["c_oracle_error"] = str(e)[:100]
    return out


def layer_report(table, launch_ms, nfw, n_img, S, f16):
    """Per-launch rates and roofline fractions from HIP-event timings.

    FLOPs come in two flavours.  ALGORITHMIC = the direct-convolution count of SURVEY.md 8(a) (what the reference's
    operators are specified to compute).  EXECUTED = what the matrix pipe actually issues: the Winograd launches run
    16/36 of the algorithmic multiplies (PMC-confirmed: SQ_VALU_MFMA_BUSY_CYCLES = executed FLOPs / 4096 x 64,
    profiles/).  A roofline FRACTION must be bounded by 1, so every `frac` here is executed work / time / peak;
    the algorithmic rate is reported beside it as `tflops_algorithmic`."""
    peak_tf = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
    layers, exec_total = [], 0.0
    for li, ((name, kern, flops, nbytes, executed), ms_sum) in enumerate(zip(table, launch_ms)):
        if f16:   # half activations and weights; the caller-side fp32 tensors of the first/last launch stay fp32
            io = 4.0 * n_img * 3 * S * S
            nbytes = (nbytes - io) / 2 + io if li in (0, len(table) - 1) else nbytes / 2
        ms = ms_sum / max(nfw, 1)
        exec_total += executed
        split = (not f16) and kern.startswith("k_conv3x3_h16")      # conv_algo "split16": this launch's MFMAs are fp16 ones (three per multiply, counted in `executed`)
        lpeak = PEAK_F16_MFMA_TFLOPS if split else peak_tf
        t_mfma, t_hbm = executed / (lpeak * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        layers.append({"layer": name, "kernel": kern, "ms": round(ms, 4),
                       "tflops_executed": round(executed / (ms * 1e-3) / 1e12, 2),
                       "tflops_algorithmic": round(flops / (ms * 1e-3) / 1e12, 2),
                       "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1), "bound": bound, "mfma_peak_tflops": lpeak,
                       "frac": round(max(t_mfma, t_hbm) / (ms * 1e-3), 4)})
    dom = max(range(len(layers)), key=lambda i: layers[i]["ms"])
    d = layers[dom]
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from separate rocprofv3 --pmc passes (profiles/README.md)
    if os.path.exists(pmc):
        try:
            tr = json.load(open(pmc))
            traffic = next((v for k, v in tr.items() if k.startswith(d["kernel"])), None)
        except Exception:
            traffic = None
    alu = None
    pmc_alu = os.path.join(ROOT, "profiles", "pmc_alu.json")   # from the `pmc_alu` pass of profiles/collect.sh (SQ_INSTS_VALU / _MFMA, MFMA busy)
    if os.path.exists(pmc_alu) and not f16:
        try:
            alu = next((v for k, v in json.load(open(pmc_alu)).items() if k.startswith(d["kernel"])), None)
        except Exception:
            alu = None
    if d["bound"] == "mfma":
        roof = {"bound": "mfma", "achieved": d["tflops_executed"], "peak": d["mfma_peak_tflops"], "unit": "TFLOP/s", "frac": d["frac"],
                "traffic": traffic, "achieved_algorithmic": d["tflops_algorithmic"]}
        if alu is not None:
            # the fp32 MFMA shares the SIMD's ALUs with every other vector instruction: (MFMA busy cycles + plain VALU instructions x 4
            # issue cycles) / SIMD cycles, from PMC counters of a profiled run of this command (profiles/README.md)
            roof["alu_frac"] = round(alu["alu_frac"], 4)
            roof["alu_note"] = (f"PMC pass: MFMA busy {alu['mfma_busy_frac']:.3f} of SIMD cycles + {alu['valu_per_mfma']:.2f} plain VALU instructions per MFMA "
                                f"x 4 cycles (x 2.8: {alu['alu_frac_c2p8']:.3f}); at the clock the part holds, not the nominal 2.4 GHz")
        if "wino" in d["kernel"]:
            roof["note"] = ("achieved = FLOPs the matrix pipe executes (Winograd F(4x2,3x3): 24/72, F(2x2,3x3): 16/36 of the direct-convolution count) / "
                            "mean launch time; achieved_algorithmic = direct-convolution FLOPs (SURVEY 8a) / the same time")
    else:
        roof = {"bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": d["frac"], "traffic": traffic}
    roof.update({"kernel": d["kernel"], "layer": d["layer"], "avg_launch_ms": d["ms"], "launches_timed": nfw,
                 "flops_per_launch": table[dom][2], "executed_flops_per_launch": table[dom][4],
                 "bytes_per_launch": table[dom][3]})
    return layers, roof, exec_total


def timed_forwards(model, x, steps, warmup):
    """`warmup` untimed + `steps` timed forwards with per-launch HIP events; -> (seconds, per-launch ms sums, forwards, last output)."""
    dev = x.device
    y = None
    for _ in range(warmup):
        y = model(x)
    torch.cuda.synchronize(dev)
    model.timing_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        y = model(x)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    launch_ms, nfw = model.timing_end()
    return elapsed, launch_ms, nfw, y


def extra_config(model, sd_default, tag, B, S, dtype, steps=20, warmup=5):
    """One of BASELINE.json's other single-GPU configs as a short leg of the default run (so the driver's record carries
    a number it timed itself): images/sec, slowest launch, parity spot check of the timed output against the CPU oracle."""
    from oracle import torch_oracle

    dev = next(model.parameters()).device
    model.compute_dtype = dtype
    try:
        x_host, clean_host, _ = synth.make_batch(B, S, S, first_index=9000)
        x = torch.from_numpy(x_host).to(dev)
        elapsed, launch_ms, nfw, y = timed_forwards(model, x, steps, warmup)
        layers, roof, exec_flops = layer_report(launch_table(B, S, S, model), launch_ms, nfw, B, S, dtype == "f16")
        ref = torch_oracle.forward(sd_default, x_host[:1]).numpy()
        got = y[:1].cpu().numpy()
        out = {"workload": tag, "batch": B, "image": [S, S, 3], "dtype": dtype, "steps": steps, "warmup": warmup,
               "images_per_sec": round(B * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
               "max_abs_err_vs_cpu_oracle": float(np.abs(got - ref).max()),
               "psnr_delta_db": abs(cid.psnr(got, clean_host[:1]) - cid.psnr(ref, clean_host[:1])),
               "slowest_launch": {k: roof[k] for k in ("layer", "kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
               "min_layer_frac": min(l["frac"] for l in layers), "layer_fracs": {l["layer"]: l["frac"] for l in layers}}
        del x, y
        return out
    finally:
        model.compute_dtype = "f32"
        model._ws = None
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=256)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--weights", default="default", choices=["default", "hot"])
    ap.add_argument("--algo", default="winograd42", choices=["winograd42", "winograd64", "direct", "split16"],
                    help="algorithm of the eight 3x3 GEMM layers: winograd42 = Winograd F(4x2,3x3) (default); winograd64 = Winograd F(2x2,3x3); "
                         "direct = 9-tap implicit GEMM (these three: exact-fp32 MFMA); split16 = OPT-IN split-operand form on the fp16 MFMA "
                         "(fp32 tensors, operands as hi + lo halfs, fp32 accumulate; the line's dtype says so) - not the headline configuration")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f32 = the reference's arithmetic (the headline metric); f16 = BASELINE configs[4] (half storage, "
                         "fp16 MFMA with fp32 accumulators) — a different numerical contract, reported for that config only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the host-pipeline and single-image-latency legs (profiling runs: only the timed batches launch kernels)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: rehearses the N>1 code path on a 1-GPU box")
    ap.add_argument("--self-launch", action="store_true",
                    help="go through the rank launcher even with --gpus 1 (rehearses launcher -> child -> relay on a 1-GPU box)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL of the N>1 code path on a box with fewer GPUs than ranks: every rank uses cuda:0 and the process group runs on gloo "
                         "(RCCL refuses two ranks on one device, so the blob travels by the fallback transport); the line is marked and is not a scaling number")
    args = ap.parse_args()

    if (args.gpus > 1 or args.self_launch) and "WORLD_SIZE" not in os.environ:
        # started as a plain command: become the launcher of N ranks (no GPU call has been made, none is made here)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    _import_runtime()

    # Everything but the final JSON line goes to stderr: RCCL prints a version banner on stdout when its first communicator is
    # created, and the contract is ONE line on stdout.  File descriptor 1 is pointed at stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", 0 if args.rehearse_shared_gpu else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, S = args.batch_per_gpu, args.size
    sd = synth.make_state_dict(args.weights)
    # rank 0 owns the checkpoint; everyone else starts from its own random init and receives the blob
    model = cid.load(sd if rank == 0 else None, device=dev, strict=True)
    model.conv_algo = args.algo
    model.compute_dtype = args.dtype
    bcast = None
    if use_dist:
        bcast = cdist.broadcast_weights_ex(model, src=0)         # the job's one collective (communicator set-up + ncclBroadcast)
        again = cdist.broadcast_weights_ex(model, src=0)         # a second one on the kept communicator: the transfer alone
        bcast["broadcast_ms_cached_comm"] = again["broadcast_ms"]

    begin, end = cdist.shard_range(B * world, rank, world)
    x_host, clean_host, noisy_host = synth.make_batch(end - begin, S, S, first_index=begin)
    x = torch.from_numpy(x_host).to(dev)
    torch.cuda.synchronize(dev)

    def barrier():
        if use_dist:
            dist.barrier()

    y = None
    for _ in range(args.warmup):
        y = model(x)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    model.timing_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = model(x)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms, nfw = model.timing_end()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_shared_gpu else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    err_all_ranks = None
    if use_dist:
        # every rank checks the first two images of ITS shard against the CPU oracle (untimed): a rank other than 0 can only pass with the
        # weights the broadcast delivered (it started from its own random init); the worst error over the ranks goes into the line
        try:
            from oracle import torch_oracle

            e = float(np.abs(y[:2].cpu().numpy() - torch_oracle.forward(sd, x_host[:2]).numpy()).max())
        except Exception:  # pragma: no cover
            e = float("nan")
        et = torch.tensor([e if e == e else 1e30], dtype=torch.float64, device=t.device)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        err_all_ranks = float(et.item())

    if rank == 0:
        table = launch_table(end - begin, S, S, model)
        f16 = args.dtype == "f16"
        peak_tf = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        layers, roof, exec_flops = layer_report(table, launch_ms, nfw, end - begin, S, f16)
        total_flops = sum(r[2] for r in table)
        res = {
            "metric": f"images/sec at batch {B}, {S}x{S}x3, {'fp16-storage' if f16 else 'fp32'} denoise forward",
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.algo != "split16" else "f32 tensors; 3x3 layers: f32 operands as hi+lo halfs on the f16 MFMA, f32 accumulate (opt-in, not the fp32 headline)",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: batch={B} per GPU, {S}x{S}x3 fp16 storage + fp16 MFMA conv-GEMM (fp32 accumulate)" if f16 else
                                    f"BASELINE configs[{1 if S == 128 else 3}]: batch={B} per GPU, {S}x{S}x3 fp32 forward, HIP conv kernels")
                                   + (f" (global batch {B * world} sharded over {world} GPUs, configs[2] shape)" if world > 1 else ""),
                       "global_batch": B * world, "image": [S, S, 3], "weights": f"synthetic seeded ({args.weights})", "conv3x3_algo": args.algo,
                       "parallelism": f"dp{world}", "inputs": "resident in HBM"},
            "whole_net_tflops_algorithmic": round(total_flops * args.steps / elapsed / 1e12, 2),
            "whole_net_tflops_executed": round(exec_flops * args.steps / elapsed / 1e12, 2),
            "whole_net_frac_of_mfma_peak": round(exec_flops * args.steps / elapsed / 1e12 / peak_tf, 4) if args.algo != "split16" else None,   # split16 mixes fp16 and fp32 MFMA launches
            "roofline": roof,
            "layers": layers,
        }
        if bcast is not None:
            # nranks = what RCCL ITSELF reports for the C-ABI communicator (cid_comm_count = ncclCommCount), not WORLD_SIZE
            res["rccl"] = {"nranks": bcast["nranks"], "transport": bcast["transport"], "broadcast_ms": round(bcast["broadcast_ms_cached_comm"], 3),
                           "first_broadcast_ms": round(bcast["broadcast_ms"], 3), "comm_setup_ms": round(bcast["setup_ms"], 2),
                           "blob_bytes": int(model.pack_weights().numel()), "collectives_in_forward": 0}
        if args.rehearse_shared_gpu:
            res["rehearsal"] = (f"{world} ranks SHARING cuda:0 over gloo: exercises launcher, sharding, barrier, max-over-ranks timing and the aggregate; "
                                "not a scaling measurement, and no RCCL (two ranks on one device are refused)")
        # parity spot-check on the timed output: 2 images vs the CPU oracle
        try:
            from oracle import torch_oracle

            ref = torch_oracle.forward(sd, x_host[:2]).numpy()
            got = y[:2].cpu().numpy()
            res["parity"] = {"max_abs_err_vs_cpu_oracle": float(np.abs(got - ref).max()),
                             "psnr_delta_db": abs(cid.psnr(got, clean_host[:2]) - cid.psnr(ref, clean_host[:2])),
                             "tolerance": "max|delta|<=1e-5, psnr_delta<=0.01 dB" if not f16 else
                                          "fp16 storage: max|delta|<=5e-4 at default weight scale (tests/test_gpu_parity.py)"}
            if err_all_ranks is not None:
                res["parity"]["max_abs_err_all_ranks"] = err_all_ranks     # ranks > 0 run on the broadcast weights
        except Exception as e:  # pragma: no cover
            res["parity"] = {"error": str(e)[:200]}
        if args.no_extras:
            emit(res)
            if use_dist:
                dist.barrier()
                cdist.WeightsComm.close_all()
                dist.destroy_process_group()
            return
        # host-buffer round trips (reported, never `value`): (1) the serial H2D -> forward -> D2H of fp32 tensors the
        # reference's callers do; (2) the same through HostPipeline (copies overlapped on separate streams);
        # (3) uint8 images in and out through HostPipeline (4x less PCIe traffic)
        try:
            from celebrity_image_denoiser_amd import HostPipeline

            xh = torch.from_numpy(x_host).pin_memory()
            yh = torch.empty(tuple(y.shape), dtype=torch.float32).pin_memory()
            yh.copy_(model(xh.to(dev, non_blocking=True)), non_blocking=True)   # untimed: first touch of the pinned buffers
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(5):
                yh.copy_(model(xh.to(dev, non_blocking=True)), non_blocking=True)
            torch.cuda.synchronize(dev)
            res["pcie_inclusive"] = {"images_per_sec": round(5 * B / (time.perf_counter() - t1), 1),
                                     "note": "per step: H2D 50 MB fp32 NCHW + forward + D2H 50 MB, pinned host buffers, one stream, rank 0 only"}
            pipe = HostPipeline(model, depth=2)
            for label, hb in (("f32_pipelined", xh), ("u8_pipelined", torch.from_numpy(noisy_host).pin_memory())):
                for _ in pipe.run([hb] * 2, copy=False):
                    pass
                t1 = time.perf_counter()
                for _ in pipe.run([hb] * 10, copy=False):
                    pass
                res["pcie_inclusive"][label + "_images_per_sec"] = round(10 * B / (time.perf_counter() - t1), 1)
            res["pcie_inclusive"]["pipelined_note"] = ("HostPipeline: upload, forward and download of consecutive batches on three HIP "
                                                       "streams; u8 = uint8 HWC images both ways (12.6 MB each way per step)")
        except Exception as e:  # pragma: no cover
            res.setdefault("pcie_inclusive", {})["error"] = str(e)[:200]
        # single-image latency, the reference server's request shape (app.py:406,433): eager (12 launches) vs one HIP-graph launch
        try:
            from celebrity_image_denoiser_amd import GraphedForward

            x1 = x[:1].contiguous()
            fast = GraphedForward(model, x1)
            for fn, key in ((lambda: model(x1), "eager_ms"), (lambda: fast(x1), "hip_graph_ms")):
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(200):
                    fn()
                torch.cuda.synchronize(dev)
                res.setdefault("latency_n1", {"shape": [1, 3, S, S]})[key] = round((time.perf_counter() - t1) / 200 * 1e3, 4)
            res["latency_n1"]["note"] = "mean over 200 back-to-back forwards of one image, host time incl. launch path; not `value`"
        except Exception as e:  # pragma: no cover
            res["latency_n1"] = {"error": str(e)[:200]}
        if world == 1 and not f16 and S == 128:
            # BASELINE.json configs[3] and configs[4] as short legs of the default run
            try:
                res["configs"] = [
                    extra_config(model, sd, "BASELINE configs[3]: batch=256 256x256x3 fp32 forward on 1 MI355X", 256, 256, "f32"),
                    extra_config(model, sd, "BASELINE configs[4]: batch=512 128x128x3 fp16 storage + fp16 MFMA conv-GEMM (fp32 accumulate)", 512, 128, "f16"),
                ]
                # fp16 contract on the He-gain ("hot") weight set too: BASELINE.md states max|delta| <= 5e-3 for this config
                from oracle import torch_oracle

                sd_hot = synth.make_state_dict("hot")
                mh = cid.load(sd_hot, device=dev, strict=True)
                mh.compute_dtype = "f16"
                xh_, _, _ = synth.make_batch(16, S, S, first_index=100)
                yh_ = mh(torch.from_numpy(xh_).to(dev)).cpu().numpy()
                res["configs"][1]["max_abs_err_hot_weights"] = float(np.abs(yh_ - torch_oracle.forward(sd_hot, xh_).numpy()).max())
                res["configs"][1]["hot_weights_images_checked"] = 16
                res["configs"][1]["frac_note"] = (
                    "layer_fracs of the MFMA-bound launches are against the nominal 2.5 PFLOP/s (2.4 GHz); this forward runs at the 1,400 W cap and "
                    "holds 1.68-1.94 GHz in them (profiles/r03_f16_pmc.md, eff. clock from SQ_BUSY_CYCLES), i.e. 1.24-1.43x these fractions of what the "
                    "part can issue at the clock it holds; up2, up1, head and last layer are HBM-bound (fraction of 8 TB/s)")
                del mh
            except Exception as e:  # pragma: no cover
                res["configs"] = {"error": str(e)[:300]}
        if world == 1 and not f16 and S == 128 and args.weights == "default":
            # (a) the same workload on the He-gain ("hot") weight set: the part runs within 2-4 % of its power cap on this path, so
            # is images/s data-dependent?  (b) the N > 1 code path at the world size this box has: RCCL communicator through the
            # C ABI + ONE ncclBroadcast of the packed blob, timed
            try:
                from oracle import torch_oracle

                sd_hot = synth.make_state_dict("hot")
                mh = cid.load(sd_hot, device=dev, strict=True)
                mh.conv_algo = args.algo
                el, lms, nf, yh2 = timed_forwards(mh, x, 5, 2)
                refh = torch_oracle.forward(sd_hot, x_host[:2]).numpy()
                res["hot_weights_leg"] = {"images_per_sec": round(B * 5 / el, 1), "ms_per_step": round(el / 5 * 1e3, 3), "steps": 5, "warmup": 2,
                                          "vs_default_weights": round((B * 5 / el) / res["value"], 4),
                                          "max_abs_err_vs_cpu_oracle": float(np.abs(yh2[:2].cpu().numpy() - refh).max()),
                                          "note": "same batch, He-gain synthetic weights (activations up to ~5, tanh to +-0.98)"}
                del mh, yh2
            except Exception as e:  # pragma: no cover
                res["hot_weights_leg"] = {"error": str(e)[:200]}
            # (c) the OPT-IN split-operand algorithm of the eight 3x3 layers (include/cid.h CID_ALGO_SPLIT16): same fp32 tensors, fp32 operands as hi + lo halfs,
            # three fp16-MFMA products per multiply, fp32 accumulation.  Reported beside the headline, never as it: `value` above is plain fp32 arithmetic.
            try:
                from oracle import torch_oracle

                legs = {}
                for wname in ("default", "hot"):
                    sdw = sd if wname == "default" else synth.make_state_dict("hot")
                    ms_ = cid.load(sdw, device=dev, strict=True)
                    ms_.conv_algo = "split16"
                    el, lms, nf, ys = timed_forwards(ms_, x, 10, 3)
                    refs = torch_oracle.forward(sdw, x_host[:4]).numpy()
                    legs[wname] = {"images_per_sec": round(B * 10 / el, 1), "ms_per_step": round(el / 10 * 1e3, 3),
                                   "max_abs_err_vs_cpu_oracle": float(np.abs(ys[:4].cpu().numpy() - refs).max()),
                                   "layers_ms": [round(v / max(nf, 1), 4) for v in lms]}
                    del ms_, ys
                res["split16_leg"] = {"conv_algo": "split16", "tail_algo": "fused", "steps": 10, "warmup": 3, "weights": legs,
                                      "vs_default_algorithm": round(legs["default"]["images_per_sec"] / res["value"], 4),
                                      "arithmetic": "fp32 tensors; 3x3 layers: operands as hi + lo halfs, three v_mfma_f32_16x16x32_f16 products per multiply, fp32 accumulate "
                                                    "(error vs float64 1.2x the exact-fp32 direct kernel's, 2-3x ATen fp32's: profiles/r04_accuracy_study.txt; parity tests at the same 1e-5) - opt-in, not the headline configuration"}
            except Exception as e:  # pragma: no cover
                res["split16_leg"] = {"error": str(e)[:300]}
            if not use_dist:
                try:
                    import socket

                    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
                    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
                    t1 = time.perf_counter()
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                    t_init = time.perf_counter() - t1
                    infos = [cdist.broadcast_weights_ex(model, src=0) for _ in range(3)]
                    y2 = model(x[:2])
                    res["dist_world1_leg"] = {"transport": infos[0]["transport"], "rccl_nranks": infos[0]["nranks"],
                                              "process_group_init_ms": round(t_init * 1e3, 2), "comm_setup_ms": round(infos[0]["setup_ms"], 2),
                                              "broadcast_ms_first": round(infos[0]["broadcast_ms"], 2),
                                              "broadcast_ms_best": round(min(i["broadcast_ms"] for i in infos), 3),
                                              "blob_bytes": int(model.pack_weights().numel()),
                                              "forward_after_broadcast_bit_equal": bool(torch.equal(y2, y[:2])),
                                              "note": "world_size 1 on this box: communicator set-up (cid_comm_*) once, then ncclBroadcast issued by "
                                                      "libcid.so on the kept communicator; the forward itself has no collective"}
                    cdist.WeightsComm.close_all()
                    dist.destroy_process_group()
                except Exception as e:  # pragma: no cover
                    res["dist_world1_leg"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd)
        emit(res)
    if use_dist:
        dist.barrier()
        cdist.WeightsComm.close_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

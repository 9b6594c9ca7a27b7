#!/usr/bin/env python3
"""Benchmark of the TATT SR training step on MI355X (BASELINE.json metric: train images/sec, whole node).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N ...        (no WORLD_SIZE in the environment: starts the line above itself as a child process, before
                                       this process has touched a GPU, relays rank 0's JSON line and exits with the child's code)

Workload = BASELINE config 2 per GPU: TSRN_TL_TRANS(STN=True, K=37) random init, train mode (dropout 0.1, BN batch
stats), fp32, batch 64 synthetic TextZoom-shaped images, frozen (precomputed) CRNN-shaped text prior,
loss = ImageLoss([1,1e-4]).mean()*100, backward, clip_grad_norm 0.25, Adam(1e-3, beta1 .5).  One step = one
forward + backward + optimiser update over one batch already resident in HBM.  N > 1 = data parallel, one process
per GPU, bucketed gradient all-reduce (AVG) over RCCL, weak scaling (64 images per GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_TRAIN = 7.70e9        # fwd+bwd, pixel+gradient loss, reference-as-executed
RPE_INPUT_PROJ_FLOP = 3 * 0.403e9    # per image as the reference executes it: W_ih x + b_ih of the RPE GRU at every step (fwd + bwd = 3x fwd);
                                     # the input is the same at every step, so this build computes it once per batch (hoisted)
RPE_SERIAL_FLOP = 3 * 0.201e9        # per image: h W_hh^T of the RPE recurrence; the second pass of a TSC step reuses the first one's chain
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: FP32 matrix peak
CONV_TRAFFIC_BYTES = 34.9e6          # profiles/r03_traffic.md (fp32 MFMA and bf16 legs; f32x3: 35.0e6)


def conv_roofline(L, B, iters=10, bf16=False, x3=False):
    """Dominant kernel = gemm_kernel<OP_IM2COL, OP_CONVW_P, FAST, TAP9, EPI>: the NHWC implicit-GEMM convolution (forward and
    dgrad) over per-step packed weights, 34 launches per training step.  Replays exactly that per-step launch mix WITH the
    epilogue variant each launch has in the step -- on [B,16,64]: 11 forward 64->64 with the BatchNorm statistics epilogue
    (EPI 1), 5 dgrad with the BatchNorm-backward + mish epilogue (EPI 2), 5 dgrad accumulating into the residual gradient and 1
    plain dgrad (EPI 0), the 64->256 up-sampling conv (EPI 0) and its dgrad (EPI 2, no activation); 10 STN-head convs (EPI 0)
    -- so that the average launch time is comparable with the rocprofv3 --stats averages of the same symbols.  Timed with
    events on the launch stream; algorithmic FLOPs per launch = 2 * pixels * Cout * KH*KW*Cin."""
    srb = (16, 64, 64, 64)
    mix = [(srb, 1, 0)] * 11 + [(srb, 2, 2)] * 5 + [(srb, 0, 1)] * 5 + [(srb, 0, 0)]          # (shape, epilogue, act | accumulate)
    mix += [((16, 64, 64, 256), 0, 0), ((16, 64, 256, 64), 2, 0)]
    stn = [(16, 32, 32, 64), (8, 16, 64, 128), (4, 8, 128, 256), (2, 4, 256, 256), (2, 2, 256, 256)]
    mix += [(c, 0, 0) for c in stn] + [((h, w, co, ci), 0, 0) for h, w, ci, co in stn]          # forward + dgrad of the STN stages 1..5
    from tatt_amd.engine import Engine
    ws = torch.empty(8 << 20, device="cuda")
    fl = 2 if bf16 else 4 if x3 else 0
    calls, flops = [], 0.0
    for (H, W, Cin, Cout), epi, opt in mix:
        R = B * H * W
        x = torch.randn(B, H, W, Cin, device="cuda")
        wf = torch.randn(Cout * 9 * Cin, device="cuda") * 0.05
        b = torch.randn(Cout, device="cuda")
        y = torch.zeros(B, H, W, Cout, device="cuda")
        if epi == 0:
            sk = Engine._conv_splitk(R, Cout, 9 * Cin)                       # the engine splits K on the small STN maps
            calls.append((L.conv_fwd_splitk, (x, wf, b, y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, opt, fl, sk, ws)))
        else:
            part = torch.empty(-(-R // 128) * 2 * Cout, device="cuda")
            bnx = torch.randn(R, Cout, device="cuda") if epi == 2 else None
            stat = torch.stack([torch.zeros(Cout), torch.ones(Cout)]).cuda() if epi == 2 else None
            ga = torch.ones(Cout, device="cuda") if epi == 2 else None
            be = torch.zeros(Cout, device="cuda") if epi == 2 else None
            calls.append((L.conv_fwd_bn, (x, wf, b if epi == 1 else None, y, B, H, W, Cin, Cout, 3, 3, Cin, Cout, 9 * Cin, 1, 0, Cin, fl, epi,
                                          bnx, Cout, stat, ga, be, opt, part)))
        flops += 2.0 * R * Cout * 9 * Cin
    for fn, c in calls:
        fn(*c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for fn, c in calls:
            fn(*c)
    e1.record()
    e1.synchronize()
    dt = e0.elapsed_time(e1) / (iters * len(calls)) * 1e-3
    fl = flops / len(calls)
    ach = fl / dt / 1e12
    if bf16:
        # With bf16 operands the same launch is bound by moving its fp32 tensors, not by the matrix core (2.5 PFLOP/s dense):
        # price it against HBM with the algorithmic bytes of an SRB-shaped launch scaled to the mix (in + out + weights, fp32).
        byts = sum(4.0 * (c[4] * c[5] * c[6] * (c[7] + c[8]) + 9 * c[7] * c[8]) for _, c in calls) / len(calls)
        return {"bound": "hbm", "kernel": "gemm_kernel<OP_IM2COL,OP_CONVW_P,FAST,BF16> (3x3 conv fwd+dgrad, per-step mix of %d launches)" % len(calls),
                "achieved": round(byts / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(byts / dt / 8e12, 4), "traffic": CONV_TRAFFIC_BYTES,
                "traffic_note": "bytes per SRB-shaped launch of gemm_halo_kernel (algorithmic 33.7e6): rocprofv3 --pmc 2*FETCH_SIZE + WRITE_SIZE, "
                                "separate passes, profiles/r03_traffic.md (not re-measured by this run)",
                "launch_us": round(dt * 1e6, 2), "bytes_per_launch": byts, "tflops": round(ach, 2), "launches_per_step": len(calls)}
    if x3:
        # three-piece split: six bf16 MFMA terms per fp32 product -- executed matrix FLOPs = 6 x algorithmic, priced against the dense bf16 peak
        return {"bound": "mfma", "kernel": "gemm_x3_kernel<OP_IM2COL,OP_CONVW_P,TAP9,EPI 0|1|2> (3x3 conv fwd+dgrad, fp32 products from three bf16 pieces "
                "per operand, per-step mix of %d launches)" % len(calls), "achieved": round(6 * ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(6 * ach / 2500.0, 4), "traffic": 35.0e6,
                "traffic_note": "bytes per SRB-shaped launch of gemm_halo_kernel<X3> (algorithmic 33.7e6), profiles/r03_traffic.md", "launch_us": round(dt * 1e6, 2), "flop_per_launch": fl,
                "algorithmic_tflops": round(ach, 2), "algorithmic_frac_of_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "launches_per_step": len(calls)}
    return {"bound": "mfma", "kernel": "gemm_kernel<OP_IM2COL,OP_CONVW_P,FAST,TAP9,EPI 0|1|2> (3x3 conv fwd+dgrad with the step's epilogue variants, per-step mix of %d launches)" % len(calls),
            "measured": "standalone replay of the step's launch mix on an otherwise idle GPU, HIP events on the launch stream",
            "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": CONV_TRAFFIC_BYTES, "traffic_note": "bytes per SRB-shaped launch (algorithmic 33.7e6): rocprofv3 --pmc 2*FETCH_SIZE"
            " + WRITE_SIZE, separate passes, profiles/r03_traffic.md (not re-measured by this run)",
            "launch_us": round(dt * 1e6, 2), "flop_per_launch": fl, "launches_per_step": len(calls)}


def useful_flop_per_image(B, tsc):
    """FLOPs this build actually performs per image (report the reduced figure when hoisted work is not done)."""
    one = FLOP_PER_IMAGE_TRAIN - RPE_INPUT_PROJ_FLOP * (B - 1) / B
    return 2 * one - RPE_SERIAL_FLOP if tsc else one


def in_step_launch_us():
    """Average duration of the dominant kernel's launches INSIDE the training step, from the rocprofv3 --kernel-trace summary of an
    eager run of this same command kept under profiles/ (the live `launch_us` above is a stand-alone replay)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_eager_summary.txt")))
    if not files:
        return None
    calls = us = 0.0
    for line in open(files[-1]):
        m = re.match(r"gemm_kernel<2, 6, true, false, true, \d(?:, false, false)?>\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            calls += float(m.group(1))
            us += float(m.group(2)) * 1e3
    return {"in_step_us": round(us / calls, 2), "in_step_launches": calls, "in_step_source": os.path.relpath(files[-1], ROOT)} if calls else None


def cpu_baseline(state, B, K, steps=5):
    """The oracle (CPU restatement of the reference step, stock torch ops) timed on this box's host cores."""
    from oracle import tatt_ref as R
    from tatt_amd.data import synthetic_batch
    hr, lr, tp = synthetic_batch(B, K, seed=123, device="cpu")
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not ("running_" in k or k.startswith("tps.") or k.endswith("pe.pe"))
             else v.clone()) for k, v in state.items()}
    st = {}
    drop = R.HashDropout(0.1, 0)
    R.train_step_config2(P, lr, hr, tp.contiguous(), st, 1, drop=drop, fused_rnn=True)      # warm-up
    ts = []
    for i in range(steps):
        t0 = time.time()
        R.train_step_config2(P, lr, hr, tp.contiguous(), st, 2 + i, drop=drop, fused_rnn=True)
        ts.append(time.time() - t0)
    dt = sorted(ts)[len(ts) // 2]
    return {"value": round(B / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {steps} training steps of the same config-2 workload (B={B}) after 1 warm-up, torch {torch.__version__} CPU"}


def launch_command(n, argv, port):
    """The torch.distributed.run command line of an N-rank run of this script (one process per GPU, one node, rendezvous on
    127.0.0.1): what `python bench.py --gpus N` starts when it was not itself started by a launcher."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """Run the N-rank job as a child process and relay it: stdout lines are passed through as they come (rank 0 prints the one
    JSON line), the exit code is the child's.  Nothing in this process has initialised a GPU (no HIP call, no
    torch.cuda.is_available()), and the child is started, not exec'd."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(n, argv, int(os.environ.get("MASTER_PORT", 0)) or free_port())
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:                 # the child ranks' stdout: rank 0's JSON line (anything else a library printed goes to stderr)
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank child run failed with exit code {rc}", file=sys.stderr, flush=True)
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--K", type=int, default=37)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full", action="store_true", help="the reference's whole train() iteration: teacher + student CRNN priors, "
                                                       "SemanticLoss, two-group Adam around the SR step (SURVEY §8f ranks 1-2)")
    ap.add_argument("--tsc", action="store_true", help="add the TSC term (second SR forward on rotated input + TRI_SSIM): "
                                                      "BASELINE config 3's step in fp32; off = config 2, the default metric")
    ap.add_argument("--bf16", action="store_true", help="matrix-core operands in bf16 (fp32 accumulate, fp32 tensors in memory): "
                                                       "with --tsc this is BASELINE config 3's arithmetic; off = fp32, the default metric")
    ap.add_argument("--f32x3", action="store_true", help="fp32 products of the trunk convolutions computed on the bf16 matrix core from "
                                                        "three bf16 pieces per operand (six terms, fp32 accumulate; error at the level "
                                                        "of the fp32 FMA chain): opt-in, the default metric uses fp32 MFMA")
    ap.add_argument("--no-stn", action="store_true", help="experiment: the network without its STN head / TPS rectification (not a BASELINE config)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} (local rank {local}) has no GPU: this node shows {ndev} device(s), the run needs {world} "
                         "(one process per GPU)")
    torch.cuda.set_device(local)
    # ONE JSON line on stdout: libraries that print to fd 1 (RCCL's version banner at communicator set-up does) get stderr instead;
    # the line itself goes through a private copy of the original stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # TATT_BENCH_FORCE_PG=1: the data-parallel code path (segmented graphs, bucketed all-reduce) through a ONE-rank RCCL group: what the
    # path itself costs on one GPU, before any link is involved (experiment; the line then carries `data_parallel`)
    use_pg = world > 1 or os.environ.get("TATT_BENCH_FORCE_PG", "0") == "1"
    if use_pg:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from tatt_amd.data import synthetic_batch
    from tatt_amd.lib import lib
    from tatt_amd.model.tsrn import TSRN_TL_TRANS
    from tatt_amd.trainer import SRTrainer

    torch.manual_seed(123)
    model = TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=not a.no_stn, mask=True, srb_nums=5, hidden_units=32,
                          text_embedding=a.K)
    state_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    model.matmul_precision = "bf16" if a.bf16 else "f32x3" if a.f32x3 else "fp32"
    kw = dict(lr=1e-3, betas=(0.5, 0.999), max_norm=0.25, loss_weight=(1.0, 1e-4), use_graph=not a.no_graph, tsc=a.tsc,
              strict_graph=True)            # a failed capture raises: no eager number under a hipGraph headline
    if a.full:
        from tatt_amd.model.crnn import CRNN
        from tatt_amd.trainer import TATTTrainer
        teacher, student = CRNN().cuda(), CRNN().cuda()
        teacher.matmul_precision = student.matmul_precision = model.matmul_precision
        trainer = TATTTrainer(model, teacher, student, lr_tpg=1e-4, **kw)
    else:
        trainer = SRTrainer(model, **kw)
    hr, lr, tp = synthetic_batch(a.batch, a.K, seed=123 + rank, device="cuda")

    for _ in range(a.warmup):
        trainer.step(lr, hr, tp)
    trainer.time_comm = use_pg                       # events around the un-overlapped part of the gradient all-reduce
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        per = trainer.step(lr, hr, tp)
        marks[i + 1].record()                        # per-step device times (median / p90 below); the metric uses the wall clock
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    dt_local = dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    dp = {}
    if use_pg:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        dts = [float(e.item()) for e in every]
        dt = max(dts)                                # MAX over ranks
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                        # the rank count as RCCL itself sees it
        # self-check: after the timed steps every replica must hold bit-identical parameters and Adam moments
        # (raises on every rank otherwise -- a diverged run must not report a throughput)
        checked = trainer.verify_replicas()
        dp = {"rccl_ranks": dist.get_world_size(), "rccl_ranks_allreduced": int(ones.item()), "replicas_verified_identical": checked,
              "backend": dist.get_backend(), "rank_ms_per_step_min": round(min(dts) / a.steps * 1e3, 3),
              "rank_ms_per_step_max": round(max(dts) / a.steps * 1e3, 3),
              "allreduce_ms_exposed": round(trainer.exposed_comm_ms(), 4), "grad_bytes_per_step": int(trainer.n) * 4,
              "buckets": [int(b - a_) * 4 for a_, b in trainer.bucket_range],
              "collective_bytes": [int(v.numel()) * 4 for st in ("srbs", "infogen") for v in trainer._bucket_views(st)]}
    trainer.check_chains()                           # the single-launch recurrences (RPE, LSTM) report timed-out waits in a flag word: fail loudly
    loss = float(per.mean().item()) * 100
    if not (loss == loss and loss < 1e6):
        raise SystemExit(f"non-finite training loss {loss}")

    if rank == 0:
        ms = dt / a.steps * 1e3
        prec = "bf16" if a.bf16 else "fp32 (3 x 3 convolutions: 3-piece bf16 split products, fp32 accumulate)" if a.f32x3 else "fp32"
        value = a.batch * world * a.steps / dt
        out = {
            "metric": "train images/sec (whole node) TATT 16x64->32x128", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_p90": round(per_step[int(0.9 * (len(per_step) - 1))], 3),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.bf16 else "bf16x3->f32" if a.f32x3 else "f32", "data": "synthetic",
            "config": {"workload": ("whole reference iteration in %s: CRNN teacher + student priors, SR step%s, SemanticLoss, "
                                    "two-group Adam, K=%d" % (prec, " + TSC" if a.tsc else "", a.K)) if a.full else
                                   ("BASELINE config 3 step in %s (config 2 + TSC: second SR forward on the rotated input, "
                                    "TRI_SSIM x10), K=%d" % (prec, a.K)) if a.tsc else
                                   "BASELINE config 2: TSRN_TL_TRANS(STN, K=%d) single training step %s, pixel+gradient loss, "
                                   "clip 0.25 + Adam, frozen text prior%s" % (a.K, prec, " -- EXPERIMENT: STN head removed, not config 2" if a.no_stn else ""),
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "lr_shape": [4, 16, 64], "hr_shape": [4, 32, 128],
                       "parallelism": f"dp{world}", "hip_graph": bool(trainer.use_graph), "loss": round(loss, 4),
                       **({"precision": "bf16 MFMA operands (round-to-nearest-even), fp32 accumulate; tensors in HBM, recurrences, "
                                        "norms, losses and optimiser fp32"} if a.bf16 else {})},
            # useful = what this build executes (RPE input projection hoisted: once per batch instead of once per step of the
            # recurrence; TSC: the second pass reuses the first pass's RPE chain); as_executed = the reference's own count
            "model_tflops": round(value * useful_flop_per_image(a.batch, a.tsc) / 1e12, 2),
            "model_frac_of_f32_mfma_peak": round(value * useful_flop_per_image(a.batch, a.tsc) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "model_gflop_per_image": round(useful_flop_per_image(a.batch, a.tsc) / 1e9, 3),
            "model_tflops_as_executed_by_reference": round(value * FLOP_PER_IMAGE_TRAIN * (2 if a.tsc else 1) / 1e12, 2),
        }
        if dp:
            out["data_parallel"] = dp
        out["roofline"] = conv_roofline(lib(), a.batch, bf16=a.bf16, x3=a.f32x3)
        if not (a.bf16 or a.f32x3):
            out["roofline"].update(in_step_launch_us() or {})
        if world == 1 and not a.no_cpu_baseline and not a.tsc and not a.full:
            out["cpu_baseline"] = cpu_baseline(state_cpu, a.batch, a.K)
        print(json.dumps(out), file=real_stdout, flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

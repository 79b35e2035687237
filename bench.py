#!/usr/bin/env python3
"""bench.py -- MPC QP-steps/sec on BASELINE.json's headline config, one process per GPU.

A "step" is one pass of the hot path over one batch: one closed-loop rollout launch
(LQ_MPC_Simulator.simulate semantics, /root/reference/utils_class.py:245-285) over
Bsz = 65536 synthetic 4-state/2-input systems with horizon N = 10 and T = 30 MPC steps, i.e.
65536 x 30 condensed box-QP solves per GPU per step, inputs already resident in HBM.

Multi-GPU: ONE global batch of `world x Bsz` distinct systems (weak scaling; config 4 is
BASELINE's fixed 262 144 systems over however many GPUs = strong scaling) is cut into contiguous
shards with lq_mpc_amd.dist.shard_batch, every rank rolls out its own shard with no data-path
collective, and the per-system closed-loop costs J_T are gathered ONCE, after the K timed
rollouts and inside the timed region, with lq_mpc_amd.dist.all_gather_costs (backend nccl = RCCL
over xGMI).  `--gather per-step` gathers after every rollout instead (overlapped with the next
one), reported under "gather".  Launch: `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`; a bare `python bench.py --gpus N` starts those N ranks itself (as child
processes, before anything touches the GPU) and relays rank 0's JSON line.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP64 matrix peak (not in the microarch guide)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def f_iter(n):
    """Algorithmic flops of one interior-point iteration: n^3/3 + 6 n^2."""
    return n ** 3 / 3.0 + 6.0 * n * n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="SURVEY 8(d) config id (3 = headline)")
    ap.add_argument("--bsz", type=int, default=0, help="systems per GPU (default: the config's)")
    ap.add_argument("--T", type=int, default=0, help="rollout length (default: the config's, 30)")
    ap.add_argument("--mode", choices=["rollout", "oneshot"], default="rollout")
    ap.add_argument("--mix", choices=["default", "hard"], default="default",
                    help="initial-state mix of the headline leg (hard: most steps constrained, lq_mpc_amd/synth.py)")
    ap.add_argument("--kernel", choices=["auto", "generic", "specialized", "workgroup"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="host threads of the cpu_baseline leg (default 16 = one GPU's share of the box's cores; 0 = every core this process may use)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (one-shot, hard mix, sweep, latency)")
    ap.add_argument("--gather", choices=["final", "per-step"], default="final",
                    help="--gpus > 1: gather J_T once after the timed rollouts (default, north_star) or after every rollout")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the sharded path on one GPU)")
    ap.add_argument("--all-on-gpu0", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--dump", default="", help="rank 0 saves the gathered J_T of the last rollout to this .npy (tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: initialise a world-size-1 process group and take the multi-rank code path (RCCL init on the "
                         "device, device-side all_gather_into_tensor, barrier, all_reduce) -- the calls an N-GPU launch makes")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process (this process
    has not touched the GPU), relay its output and exit with its code."""
    import torch
    # (device_count() may bring up the HIP runtime in THIS process on some builds; the ranks are fresh child processes, which is
    # allowed -- but never run this launcher under rocprofv3: tools/prof.sh refuses --gpus > 1)
    have = torch.cuda.device_count()
    if not args.all_on_gpu0 and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    from lq_mpc_amd import BatchSolver, synth
    from lq_mpc_amd import dist as ld

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    dist = None
    if args.all_on_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or args.force_dist          # the collective path (a world of one rank still runs every call of it)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:              # (--force-dist without a launcher)
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where the collectives run

    cfg = synth.CONFIGS[args.config]
    nx, nu, N = cfg["nx"], cfg["nu"], cfg["N"]
    n = N * nu
    T = args.T or cfg["T"]
    strong = args.config == 4 and not args.bsz      # BASELINE config 4: 262 144 systems in total, sharded over the GPUs
    gold = os.path.join(ROOT, "tests", "golden")
    # ONE global batch of distinct systems, cut into contiguous shards (lq_mpc_amd/dist.py).  Weak scaling: world x (the N = 1
    # batch size), the same generator and difficulty mix, so that the driver's efficiency figure measures the parallel
    # overheads and not a change of workload.
    Bglobal = cfg["Bsz"] if strong else world * (args.bsz or cfg["Bsz"])
    gb = synth.make_batch(args.config, Bsz=Bglobal, fixture_dir=gold, mix=args.mix)
    b = ld.shard_batch(gb, rank, world) if distributed else gb
    Bsz = b["A"].shape[-1]
    lo = b["shard"][0] if distributed else 0
    del gb

    dA = torch.from_numpy(b["A"]).to(dev); dB = torch.from_numpy(b["B"]).to(dev); dx0 = torch.from_numpy(b["x0"]).to(dev)
    dJT = torch.empty(Bsz, dtype=torch.float64, device=dev)
    du0 = torch.empty((nu, Bsz), dtype=torch.float64, device=dev)
    dVN = torch.empty(Bsz, dtype=torch.float64, device=dev)
    dst = torch.empty(Bsz, dtype=torch.int32, device=dev)
    dit = torch.empty(Bsz, dtype=torch.int32, device=dev)

    stream = torch.cuda.current_stream(dev).cuda_stream
    kern = {"auto": 0, "generic": 1, "specialized": 2, "workgroup": 3}[args.kernel]
    s = BatchSolver(local_rank, stream=stream, kernel=kern)
    s.reserve(nx, nu, N, Bsz, T)
    shared = (b["Q"], b["R"], b["P"], b["lb"], b["ub"])

    def launch_rollout(x0=dx0):
        s.rollout_batch_dev(nx, nu, N, Bsz, T, dA, dB, *shared, x0, b["A_true"], b["B_true"], dJT, dstatus=dst, diters=dit)

    def launch_oneshot():
        s.solve_batch_dev(nx, nu, N, Bsz, dA, dB, *shared, dx0, du0, dVN, dstatus=dst, diters=dit)

    # ---- the exchange step of the path: the final cost curves ----
    gathered = {"J": None, "pending": None, "k": 0, "ms": None}

    def gather_final():
        src = dJT if cdev.type == "cuda" else dJT.cpu()              # gloo rehearsal: through host memory
        gathered["J"] = ld.all_gather_costs(src, Bglobal)

    if distributed and args.gather == "per-step":
        # the collective of step k runs while the rollout of step k+1 computes: J_T is copied to one of two staging buffers on the
        # launch stream, gathered asynchronously from it and waited for one step later (and before the timed region closes)
        sizes = [ld.shard_bounds(Bglobal, r, world) for r in range(world)]
        m = max(hi - lo_ for lo_, hi in sizes)
        stage_bufs = [torch.zeros(m, dtype=torch.float64, device=cdev) for _ in range(2)]
        out_bufs = [torch.empty(world * m, dtype=torch.float64, device=cdev) for _ in range(2)]

    def gather_step():
        k = gathered["k"] & 1
        stage_bufs[k][:Bsz].copy_(dJT)                               # device-to-device under nccl, device-to-host under gloo
        if gathered["pending"] is not None:
            gathered["pending"].wait()
        gathered["pending"] = dist.all_gather_into_tensor(out_bufs[k], stage_bufs[k], async_op=True)
        gathered["k"] += 1

    def gather_flush():
        if gathered["pending"] is not None:
            gathered["pending"].wait()
            gathered["pending"] = None

    def run(launch, steps, warmup, qp_per_launch, with_gather=True):
        per_step = distributed and with_gather and args.gather == "per-step"
        final = distributed and with_gather and args.gather == "final"
        for _ in range(warmup):
            launch()
            if per_step:
                gather_step()
        if per_step:
            gather_flush()
        if final:
            gather_final()                                           # warm the collective up as well
        # one pair of HIP events around the K launches, on the launch stream (a pair per launch puts two marker packets
        # between consecutive launches and costs ~8 % of a 0.47 ms step); recorded BEFORE the gather, so that the kernel
        # time of the roofline holds no collective
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            launch()
            if per_step:
                gather_step()
        ev1.record()
        if per_step:
            gather_flush()
        if final:
            tg = time.perf_counter()
            gather_final()
            torch.cuda.synchronize()
            gathered["ms"] = (time.perf_counter() - tg) * 1e3        # includes waiting for this rank's last rollout
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        kernel_ms = ev0.elapsed_time(ev1) / steps
        return dt, kernel_ms, qp_per_launch * steps / dt

    def totals():
        """(sum of iters, instances with non-zero status) over all ranks."""
        agg = torch.tensor([float(dit.double().sum().item()), float((dst != 0).sum().item())], dtype=torch.float64, device=cdev)
        if distributed:
            dist.all_reduce(agg)
        return float(agg[0].item()), int(agg[1].item())

    launch = launch_rollout if args.mode == "rollout" else launch_oneshot
    per_inst = T if args.mode == "rollout" else 1
    qp_local = Bsz * per_inst                      # this rank's QP-steps per launch (roofline of this rank's kernel)
    qp_global = Bglobal * per_inst                 # all ranks' (value)
    dt, kernel_ms, value = run(launch, args.steps, args.warmup, qp_global)
    kernel_name = s.last_kernel()
    iters_total, status_bad = totals()
    iters_mean = iters_total / qp_global

    gather_info = None
    if distributed:
        gather_info = {"mode": args.gather, "world_size": world, "collective": "all_gather_into_tensor (lq_mpc_amd.dist.all_gather_costs)",
                       "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else " (rehearsal)"),
                       "bytes_per_rank": 8 * Bsz}
        if args.gather == "final":
            J = gathered["J"]
            mine = J[lo:lo + Bsz].to(dev)
            gather_info["ms_final_gather_rank0"] = round(gathered["ms"], 4)
            gather_info["own_shard_intact"] = bool(torch.equal(mine, dJT))
            gather_info["J_T_sum"] = float(J.double().sum().item())
            if args.dump and rank == 0:
                np.save(args.dump, J.cpu().numpy())
        else:
            k = (gathered["k"] - 1) & 1                  # the last per-step gather (flushed before the timed region closed)
            J = out_bufs[k]
            m_ = stage_bufs[k].numel()
            mine = J[rank * m_: rank * m_ + Bsz].to(dev)
            gather_info["own_shard_intact"] = bool(torch.equal(mine, dJT))
            if args.dump and rank == 0:
                sizes_ = [ld.shard_bounds(Bglobal, r_, world) for r_ in range(world)]
                np.save(args.dump, torch.cat([J[r_ * m_: r_ * m_ + (hi_ - lo_)] for r_, (lo_, hi_) in enumerate(sizes_)]).cpu().numpy())
    elif args.dump:
        np.save(args.dump, dJT.cpu().numpy())

    # ---- roofline of the dominant kernel (the one launch per step), per launch, on rank 0's shard ----
    def roofline_of(kernel_ms, iters_mean, mode, qp_launch):
        per = T if mode == "rollout" else 1
        cond_flops = (N * nx * n * n + 2 * N * nx * nx * nu)            # condensing, once per instance
        f_step = iters_mean * f_iter(n) + 2 * n * nx + cond_flops / per
        alg_bytes = 8 * (nx * nx + nx * nu + nx + (1 if mode == "rollout" else nu + 1))   # per instance per launch
        tflops = f_step * qp_launch / (kernel_ms * 1e-3) / 1e12
        gbs = alg_bytes * (qp_launch / per) / (kernel_ms * 1e-3) / 1e9
        return f_step, alg_bytes, tflops, gbs

    def f_step_no_setup(iters_mean):
        """SURVEY 8(d)'s rollout formula read literally: iters * F_it(n) + 2 n n_x, no condensing term."""
        return iters_mean * f_iter(n) + 2 * n * nx

    f_step, alg_bytes, tflops, gbs = roofline_of(kernel_ms, iters_mean, args.mode, qp_local)
    # HBM traffic per launch: the PMC counters cannot be read from inside this process; the figure is the committed rocprofv3
    # --pmc pass of THIS command line (profiles/pmc_traffic.json, produced by tools/prof_summary.py) and is reported only when that
    # pass's kernel time agrees with the live one to 15 % -- otherwise null (stale profile)
    traffic, traffic_source, mfma_insts = None, None, None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            rec = json.load(open(tf)).get(f"{kernel_name}:{args.mode}:C{args.config}:Bsz{Bsz}:T{T}:{args.mix}")
            if rec:
                ref_ms = rec.get("kernel_ms_per_launch")
                if ref_ms and abs(ref_ms - kernel_ms) <= 0.15 * kernel_ms:
                    traffic = rec["hbm_bytes_per_launch"]
                    mfma_insts = rec.get("mfma_f64_insts_per_launch")
                    traffic_source = f"profiles/pmc_traffic.json ({rec.get('source', '?')}, kernel {ref_ms} ms there)"
                else:
                    traffic_source = f"profiles/pmc_traffic.json entry is stale (kernel {ref_ms} ms there, {kernel_ms:.4f} ms now): not reported"
        except Exception as e:                                          # a malformed file must not cost the bench line
            traffic_source = f"profiles/pmc_traffic.json unreadable: {e}"
    wg = "wg" in kernel_name
    roofline = {
        "kernel": kernel_name, "bound": "mfma",
        "unit_executing": "mfma_f64 16x16x4 (block products) + valu_fp64" if wg else
                          ("mfma_f64 4x4x4 (set-up: Riccati recursion, W, P, G) + valu_fp64 / DPP (active-set iterations, closed loop)"
                           if "r16" in kernel_name or "r64" in kernel_name else "valu_fp64"),
        "bound_detail": "fp64 compute roof: 78.6 TFLOP/s dense, the same for the vector ALU and v_mfma_f64 ('mfma' is the contract's "
                        "name for the compute roof; unit_executing says which unit runs this kernel)",
        "achieved": round(tflops, 4), "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": round(tflops / FP64_PEAK_TFLOPS, 5),
        "frac_no_setup": round(f_step_no_setup(iters_mean) * qp_local / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5),
        "flops_per_qp_step_no_setup": round(f_step_no_setup(iters_mean), 1),
        "traffic": traffic, "traffic_source": traffic_source,
        "kernel_ms_per_launch": round(kernel_ms, 4), "iters_mean": round(iters_mean, 3),
        "flops_per_qp_step": round(f_step, 1), "alg_bytes_per_instance": alg_bytes,
        "hbm": {"achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 7)},
        "note": "the fp64 compute roof binds (SURVEY 8(d)): ~250 B of unique HBM traffic vs ~40 kflop per QP-step; "
                "flops = iters_mean*(n^3/3+6n^2)+2*n*nx+condensing/T (frac) or without the condensing term (frac_no_setup: SURVEY 8(d)'s "
                "rollout formula read literally); iters_mean = KKT factorisations per QP-step "
                "(interior-point + active-set iterations; 0 for steps the presolve finishes); kernel time = HIP events around "
                "the K launches of this rank, collectives excluded",
    }
    if mfma_insts:      # the share of the arithmetic that runs on the matrix cores (instruction count from the committed profile)
        roofline["mfma"] = {"achieved": round(mfma_insts * 2048 / (kernel_ms * 1e-3) / 1e12, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "insts_per_launch": mfma_insts}

    # ---- extra legs (N = 1 only): one-shot solves, the hard mix, the fused sweep, small-batch latency ----
    extra = {}
    if world == 1 and not args.no_extras and args.mode == "rollout":
        dt1, kms1, v1 = run(launch_oneshot, max(args.steps, 20), 3, Bsz)
        it1, _ = totals()
        extra["oneshot"] = {"value": round(v1, 1), "unit": "QP-steps/s", "kernel_ms_per_launch": round(kms1, 4),
                            "iters_mean": round(it1 / Bsz, 3), "qp_per_launch": Bsz}
        if args.mix == "default" and args.config != 1:
            # hard mix: same models, initial states 6-24x outside the region where the box is inactive -> most steps are
            # constrained QPs; the presolve is not what this leg measures
            hb = synth.make_batch(args.config, Bsz=Bsz, fixture_dir=gold, mix="hard")
            dxh = torch.from_numpy(hb["x0"]).to(dev)
            dth, kmsh, vh = run(lambda: launch_rollout(dxh), max(args.steps // 2, 5), 2, qp_local)
            ith, bad_h = totals()
            fs, _, tfl, _ = roofline_of(kmsh, ith / qp_local, "rollout", qp_local)
            m = min(Bsz, 256)
            idx = np.arange(m)
            tr = s.rollout_batch(T, N, hb["A"][:, :, :m].copy(), hb["B"][:, :, :m].copy(), *shared, hb["x0"][:, :m].copy(),
                                 hb["A_true"], hb["B_true"], want_traj=True)
            extra["rollout_hard"] = {
                "value": round(vh, 1), "unit": "QP-steps/s", "kernel_ms_per_launch": round(kmsh, 4),
                "iters_mean": round(ith / qp_local, 3), "status_nonzero": bad_h,
                "constrained_step_share": round(synth.constrained_share(hb, tr["X"], idx), 4),
                "constrained_step_share_sample": f"first {m} instances, all {T} steps, unconstrained minimiser outside the box",
                "flops_per_qp_step": round(fs, 1), "roofline_frac": round(tfl / FP64_PEAK_TFLOPS, 5),
                "workload": f"C{args.config} shapes and models, x0 scaled so the LQR input at step 0 is s*u_max, s ~ U{list(synth.HARD_S)}"}
            del dxh
        if n <= 32 and args.config != 1:
            # the reference's sweep at this shape: M_V over K = 8 points of the level set x'Qx = r^2 + the rollout, one fused launch
            from lq_mpc_amd import sweep as sw
            x0s = sw.circle_generator(8, 1.0, float(np.median(np.einsum("ib,ij,jb->b", b["x0"], b["Q"], b["x0"]))), b["Q"])
            dMV = torch.empty(Bsz, dtype=torch.float64, device=dev)

            def launch_sweep():
                s.sweep_batch_dev(nx, nu, N, Bsz, T, dA, dB, *shared, dx0, x0s, b["A_true"], b["B_true"], dJT, dMV, dstatus=dst, diters=dit)
            dts, kmss, vs = run(launch_sweep, max(args.steps // 2, 5), 2, Bsz * (T + 8))
            its, bad_s = totals()
            extra["sweep"] = {"value": round(vs, 1), "unit": "QP-steps/s", "kernel_ms_per_launch": round(kmss, 4),
                              "qp_per_launch": Bsz * (T + 8), "iters_mean": round(its / (Bsz * (T + 8)), 3), "status_nonzero": bad_s,
                              "kernel": s.last_kernel(),
                              "workload": "lqmpc_sweep_batch_dev: max V_N over 8 level-set points + the T-step rollout per system "
                                          "(utils_class.py:813-833), one launch"}
        # the rest of data_generation per system (utils_class.py:837-859): dlqr + energy_decreasing + energy_bound on the GPU
        if args.config != 1:
            dMV = torch.full((Bsz,), 0.5, dtype=torch.float64, device=dev)
            dlev = torch.full((Bsz,), 5e-3, dtype=torch.float64, device=dev)
            outs = [torch.empty(Bsz, dtype=torch.float64, device=dev) for _ in range(5)]
            dstb = torch.empty(Bsz, dtype=torch.int32, device=dev)
            xb, pb = np.ascontiguousarray(b["x0"][:, 0]), np.array([0.1, 1.0, 0.6])

            def launch_bounds():
                s.bounds_batch_dev(nx, nu, N, Bsz, dA, dB, b["Q"], b["R"], b["lb"], b["ub"], dlev, dlev, dMV, xb, pb, 1.0,
                                   dalpha=outs[0], dbeta=outs[1], dxi=outs[2], deta=outs[3], dbound=outs[4], dstatus=dstb)
            for _ in range(2):
                launch_bounds()
            s.timer_begin()
            for _ in range(3):
                launch_bounds()
            msb = s.timer_end() / 3
            extra["bounds"] = {"value": round(Bsz / (msb * 1e-3), 1), "unit": "systems/s", "kernel_ms_per_launch": round(msb, 4),
                               "dlqr_not_converged": int((dstb == 1).sum().item()) + int((dstb == 2).sum().item()),
                               "kernel": s.last_kernel(),
                               "workload": "lqmpc_bounds_batch_dev: dlqr (doubling) + local radius + stability numbers + alpha/beta/xi/eta/bound "
                                           "per system (utils_class.py:837-859); on chip: n_x x n_x work on the matrix core (four systems per "
                                           "wavefront), Gamma'Gamma from the Toeplitz form, Householder tridiagonalisation by rows in LDS, "
                                           "extreme eigenvalues by multisection on the Sturm count"}
        if args.config != 1:
            # the engine north_star names, alone: Mehrotra interior point + active-set polish, no warm start, no presolve shortcut
            # beyond the default (options.warm_start = 0)
            s.set_options(warm_start=0)
            try:
                dti, kmsi, vi = run(launch_rollout, max(args.steps // 4, 3), 1, qp_local)
                iti, bad_i = totals()
            finally:
                s.set_options(warm_start=-1)
            fsi, _, tfli, _ = roofline_of(kmsi, iti / qp_local, "rollout", qp_local)
            extra["ipm_only"] = {"value": round(vi, 1), "unit": "QP-steps/s", "kernel_ms_per_launch": round(kmsi, 4),
                                 "iters_mean": round(iti / qp_local, 3), "status_nonzero": bad_i, "kernel": s.last_kernel(),
                                 "flops_per_qp_step": round(fsi, 1), "roofline_frac": round(tfli / FP64_PEAK_TFLOPS, 5),
                                 "workload": "the headline rollout with options.warm_start = 0: every QP the presolve does not finish goes "
                                             "through the primal-dual interior-point loop + exact polish"}
        # PCIe: the headline batch's inputs host -> device and its result back (pinned host memory), never part of `value`
        hA, hB, hx = (torch.from_numpy(b[k]).pin_memory() for k in ("A", "B", "x0"))
        hJ = torch.empty(Bsz, dtype=torch.float64).pin_memory()
        tA, tB, tx = torch.empty_like(dA), torch.empty_like(dB), torch.empty_like(dx0)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        h2d, d2h = [], []
        for _ in range(4):
            e0.record(); tA.copy_(hA, non_blocking=True); tB.copy_(hB, non_blocking=True); tx.copy_(hx, non_blocking=True); e1.record()
            hJ.copy_(dJT, non_blocking=True); e2.record(); torch.cuda.synchronize()
            h2d.append(e0.elapsed_time(e1)); d2h.append(e1.elapsed_time(e2))
        in_bytes = 8 * (tA.numel() + tB.numel() + tx.numel())
        h2d_ms, d2h_ms = min(h2d), min(d2h)
        extra["pcie"] = {"h2d_ms": round(h2d_ms, 4), "d2h_ms": round(d2h_ms, 4), "h2d_bytes": in_bytes, "d2h_bytes": 8 * Bsz,
                         "h2d_GBps": round(in_bytes / (h2d_ms * 1e-3) / 1e9, 2),
                         "value_pcie_inclusive": round(qp_local / ((h2d_ms + kernel_ms + d2h_ms) * 1e-3), 1),
                         "note": "A, B, x0 of the headline batch from pinned host memory, J_T back; HIP events, best of 4; "
                                 "value_pcie_inclusive = QP-steps / (h2d + one launch + d2h), reported beside `value`, never as it"}
        del tA, tB, tx
        # a shape WITHOUT a prebuilt instantiation beside its tabled neighbour: the 16-lane-row kernel compiled at run time
        # (lqmpc_jit.hip; rounds 1-2 dropped such shapes onto the generic kernel, 650x slower)
        if args.config in (3, 4):
            Nj = N + 2
            try:
                t0 = time.perf_counter()
                s.rollout_batch_dev(nx, nu, Nj, Bsz, T, dA, dB, *shared, dx0, b["A_true"], b["B_true"], dJT, dstatus=dst, diters=dit)
                torch.cuda.synchronize()
                first_s = time.perf_counter() - t0
                dtj, kmsj, vj = run(lambda: s.rollout_batch_dev(nx, nu, Nj, Bsz, T, dA, dB, *shared, dx0, b["A_true"], b["B_true"], dJT,
                                                                dstatus=dst, diters=dit), max(args.steps // 2, 5), 2, qp_local)
                itj, bad_j = totals()
                extra["untabled_shape"] = {"shape": [nx, nu, Nj], "value": round(vj, 1), "unit": "QP-steps/s",
                                           "kernel_ms_per_launch": round(kmsj, 4), "kernel": s.last_kernel(), "iters_mean": round(itj / qp_local, 3),
                                           "status_nonzero": bad_j, "first_call_s": round(first_s, 3),
                                           "tabled_neighbour": {"shape": [nx, nu, N], "kernel_ms_per_launch": round(kernel_ms, 4)},
                                           "ratio_to_tabled_neighbour": round(kmsj / kernel_ms, 3),
                                           "workload": "the headline batch with horizon N + 2 (no prebuilt instantiation): run-time compiled "
                                                       "16-lane-row kernel; first_call_s includes the compile unless the code object was cached"}
            except Exception as e:
                extra["untabled_shape"] = {"error": repr(e)}
        # closed-loop rollouts of small batches: us per MPC step (the reference runs ONE system per simulate(), utils_class.py:245-285)
        small = {}
        for m in (1, 64, 4096):
            if m > Bsz:
                continue
            mA, mB, mx = (torch.from_numpy(np.ascontiguousarray(b[k][..., :m])).to(dev) for k in ("A", "B", "x0"))
            mJ = torch.empty(m, dtype=torch.float64, device=dev)
            for _ in range(3):
                s.rollout_batch_dev(nx, nu, N, m, T, mA, mB, *shared, mx, b["A_true"], b["B_true"], mJ)
            s.timer_begin()
            for _ in range(20):
                s.rollout_batch_dev(nx, nu, N, m, T, mA, mB, *shared, mx, b["A_true"], b["B_true"], mJ)
            us = s.timer_end() / 20 * 1e3
            small[str(m)] = {"rollout_us": round(us, 1), "us_per_mpc_step": round(us / T, 3), "kernel": s.last_kernel()}
        extra["rollout_small_batches"] = {"by_batch": small, "T": T, "note": "lqmpc_rollout_batch_dev on the first m systems, resident "
                                          "data, back-to-back launches, HIP events; one launch = T dependent MPC steps"}
        # the reference's own workload end to end: LQ_RDP_Behavior_Multiple.data_generation (utils_class.py:766-959), 57 001 QPs +
        # the bound coefficients of 1 500 systems, through the host API (numpy in / out, 6 sweep + 6 bounds calls + 2 small ones)
        try:
            from lq_mpc_amd.sweep import LQ_RDP_Behavior_Multiple
            info_opc = {"A": synth.A_REF, "B": synth.B_REF, "Q": 2.0 * np.eye(2), "R": np.eye(1), "F_u": np.array([[10.0], [-10.0]])}
            info_N = {"N_min": 6, "N_max": 10, "N_nominal": 7, "N_opc": 30, "N_mpc": 30}
            info_e = {"e_min": 1e-3, "e_max": 1e-2, "e_nominal": 5e-3}
            info_ref = {"x_ref": np.zeros((2, 7)), "u_ref": np.zeros((1, 7)), "x_ref_long": np.zeros((2, 30)), "u_ref_long": np.zeros((1, 30))}
            sdg = BatchSolver(local_rank)
            beh = LQ_RDP_Behavior_Multiple(info_opc, info_N, info_e, 20, "f", data_dir=gold, solver=sdg)
            pdg = np.array([0.1, 1.0, 0.6])
            beh.data_generation(8, 1.5, info_ref, pdg); beh.data_generation(8, 1.5, info_ref, pdg, concurrent=False)
            tw, tseq = [], []
            for _ in range(5):
                t0 = time.perf_counter(); og = beh.data_generation(8, 1.5, info_ref, pdg); tw.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); beh.data_generation(8, 1.5, info_ref, pdg, concurrent=False); tseq.append(time.perf_counter() - t0)
            ref_npz = np.load(os.path.join(gold, "data_lq_mpc_multipleSys.npz"))
            dev_ = float(np.max(np.abs(og["true_cost_error"] - ref_npz["true_cost_error"]) / np.abs(ref_npz["true_cost_error"])))
            sdg.close()
            extra["data_generation"] = {"wall_ms": round(min(tw) * 1e3, 3), "wall_ms_median": round(sorted(tw)[len(tw) // 2] * 1e3, 3),
                                        "wall_ms_one_handle": round(min(tseq) * 1e3, 3),
                                        "qp_solves": 57001, "value": round(57001 / min(tw), 1), "unit": "QP-steps/s",
                                        "true_cost_error_max_rel_dev_vs_reference_npz": dev_,
                                        "workload": "lq_mpc_amd.sweep.LQ_RDP_Behavior_Multiple.data_generation on the reference's inputs "
                                                    "(error_{A,B}_f.npy, working_example_multiple.py constants): 1 500 systems x (8 open-loop "
                                                    "+ 30 closed-loop QPs) + V_expert + the 13 npz arrays, host API, wall clock; wall_ms: the six passes (error levels, "
                                                    "five horizons) on six handles / streams from a thread pool, wall_ms_one_handle: one after the other"}
        except Exception as e:                                          # (never at the price of the bench line)
            extra["data_generation"] = {"error": repr(e)}
        # single-call latency: the reference's call shape is ONE instance per solve() (utils_class.py:269)
        lat = {}
        for m in (1, 64, 4096):
            if m > Bsz:
                continue
            hA, hB, hx = b["A"][:, :, :m].copy(), b["B"][:, :, :m].copy(), b["x0"][:, :m].copy()
            for _ in range(3):
                s.solve_batch(N, hA, hB, *shared, hx)
            t0 = time.perf_counter()
            reps = 50
            for _ in range(reps):
                s.solve_batch(N, hA, hB, *shared, hx)
            host_us = (time.perf_counter() - t0) / reps * 1e6
            mA, mB, mx = (torch.from_numpy(a).to(dev) for a in (hA, hB, hx))
            for _ in range(3):
                s.solve_batch_dev(nx, nu, N, m, mA, mB, *shared, mx, du0, dVN, dstatus=dst, diters=dit)
            s.timer_begin()
            for _ in range(reps):
                s.solve_batch_dev(nx, nu, N, m, mA, mB, *shared, mx, du0, dVN, dstatus=dst, diters=dit)
            dev_us = s.timer_end() / reps * 1e3
            lat[str(m)] = {"solve_batch_host_us": round(host_us, 1), "solve_batch_dev_us": round(dev_us, 1)}
        extra["latency"] = {"per_call_us_by_batch": lat,
                            "note": "one-shot solves of the first m systems.  solve_batch: numpy in/out incl. H2D/D2H and the final "
                                    "sync (what LQ_MPC_Controller.solve costs per call at m = 1); solve_batch_dev: back-to-back "
                                    "launches on resident data, HIP events"}

    # ---- CPU baseline: the oracle (port) at 1 thread, --cpu-threads and all host cores; bounded samples; rank 0 at N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        native = orc.use_native_build()                    # gcc -O3 -march=native on THIS host (falls back to the shipped x86-64-v3 build)
        avail = len(os.sched_getaffinity(0))

        def cpu_run(m, threads):
            A = np.ascontiguousarray(b["A"][:, :, :m]); Bm = np.ascontiguousarray(b["B"][:, :, :m])
            x0 = np.ascontiguousarray(b["x0"][:, :m])
            t = time.perf_counter()
            if args.mode == "rollout":
                orc.rollout_batch(T, N, A, Bm, *shared, x0, b["A_true"], b["B_true"], threads=threads)
                q = m * T
            else:
                orc.solve_batch(N, A, Bm, *shared, x0, threads=threads)
                q = m
            return q, time.perf_counter() - t

        def cpu_leg(threads, wall=1.0):
            m0 = int(min(Bsz, max(16, 16 * threads)))
            q0, t0 = cpu_run(m0, threads)
            m = int(min(Bsz, max(m0, m0 * wall / max(t0, 1e-4))))   # ~`wall` seconds per pass when the batch allows
            q, t, reps = 0, 0.0, 0
            while t < wall and reps < 50:
                qi, ti = cpu_run(m, threads)
                q += qi; t += ti; reps += 1
            return {"value": round(q / t, 1), "unit": "QP-steps/s", "cores": threads,
                    "sample": f"first {m} instances of the same batch x {reps} passes ({q} QP-steps, {t:.2f} s wall, {t * threads:.0f} core-seconds)"}
        legs = {}
        for th in sorted({1, max(1, min(avail, args.cpu_threads or avail)), avail}):
            legs[th] = cpu_leg(th, wall=2.0 if th == 1 else 1.0)
        best = max(legs, key=lambda k: legs[k]["value"])      # (a shared box: "all cores" can be slower than one GPU's share of them)
        top = legs[best]
        cpu = {"value": top["value"], "unit": "QP-steps/s", "cores": best, "kind": "port",
               "nproc": os.cpu_count(), "cores_available_to_process": avail,
               "build": "gcc -O3 -march=native -fopenmp on this host" if native else "shipped build (-march=x86-64-v3): no compiler on this host",
               "single_thread": legs[1], "all_cores": legs[avail], "by_threads": {str(k): v for k, v in legs.items()},
               "sample": top["sample"] + "; exact active-set oracle (oracle/lqmpc_oracle.c), OpenMP over instances; `value` = the best of "
                         "the thread counts tried (1, --cpu-threads, every core this process may use: by_threads), `cores` = its threads; "
                         "the reference's cvxpy path is not measurable in this pipeline (cvxpy absent, SURVEY 8(d))"}

    if rank == 0:
        per_gpu = f"{Bglobal} systems over {world} GPU(s)" if strong else f"{Bsz} systems/GPU"
        out = {
            "metric": "MPC QP-steps/sec (batched systems) at n_x=4,n_u=2,N=10; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "QP-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C{args.config}: {per_gpu}, n_x={nx}, n_u={nu}, N={N}, box |u|<=0.1, "
                                   + (f"closed-loop rollout T={T} (one launch = {Bsz}x{T} QP-steps)" if args.mode == "rollout"
                                      else "one-shot open-loop solve (one launch = one QP per system)")
                                   + ("" if args.mix == "default" else f", {args.mix} initial-state mix"),
                       "mode": args.mode, "batch_per_gpu": Bsz, "batch_global": Bglobal, "T": per_inst, "mix": args.mix,
                       "parallelism": (f"dp{world}: one global batch of {Bglobal} distinct systems in contiguous shards "
                                       f"(lq_mpc_amd.dist.shard_batch), no data-path collective, J_T gathered "
                                       + ("once after the timed rollouts" if args.gather == "final" else "after every rollout (overlapped)"))
                       if world > 1 else "dp1",
                       "options": s.get_options(), "status_nonzero": status_bad},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if gather_info:
            out["gather"] = gather_info
        out.update(extra)
        print(json.dumps(out), flush=True)
    s.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: exact-GP fits/sec (and predict points/sec) at N=4096, d=8, fp64 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = every one of the rank's CELLS_PER_STEP independent cells fitted once (F1) at fixed
hyperparameters by ONE batched launch sequence (gprx_factorize_batch: the cell index rides in every launch's
grid); per cell: stationary-kernel matrix build (lower tiles) + blocked fp64-MFMA Cholesky with y carried as an
extra row + backward solve for alpha + log marginal likelihood returned to the host -- BASELINE.json configs[1]
("Single cell, N=4096 d=8 RBF fp64 on 1 MI355X: HIP kernel build + blocked MFMA Cholesky").  The cells are the
reference's own unit of independent work: the per-mode models of one GPRAS (gpr.py:272-274) share x and differ in y
and hyperparameters; here every cell has its own y column and its own hyperparameters, so nothing is shared between
cells but the inputs x.  Inputs are resident in HBM when the timed region starts.  With N > 1 ranks every rank fits its own cells (independent units, SURVEY.md
section 8e: no data-path collective) and one RCCL all_gather collects the results at the end; the
reported value is all ranks' fits divided by the slowest rank's time ("weak" scaling).

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      -- the dominant kernel (the Cholesky trailing-update GEMM) against the fp64 MFMA peak,
                   per-launch durations taken with HIP events on the launch stream in an instrumented
                   pass over the same workload;
  cpu_baseline  -- the numpy/scipy oracle (a restatement, "port": the reference cannot be imported here)
                   timed on this box's host cores on a bounded sample;
  extra         -- F2 (objective + gradient), F3 (50 L-BFGS-B iterations, Matern-5/2 ARD) and predict rates.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_TRAIN, DIM, N_TEST = 4096, 8, 100_000
CELLS_PER_STEP = 256  # cells per batched launch sequence per GPU per step (round 3, one box: 128 -> 2138, 192 -> 2157, 256 -> 2168, 384 -> 2173 fits/s;
                      # round 1: 16 -> 1600, 64 -> 1840, 96 -> 1880, 128 -> 1896); 256 cells = 35 GB of the 288 GB
FP64_MFMA_PEAK_TFLOPS = 78.6  # 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz: half the f32 matrix rate of MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


MAIN_KERNEL = "gemm_f64_kernel<0,1,64,64,0,0,1>"
PMC_FILE = "r04_pmc_hbm_traffic.json"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be read
    inside this process); None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=CELLS_PER_STEP, help="independent cells per GPU per step")
    ap.add_argument("--no-extras", action="store_true", help="skip F2/F3/predict/cpu legs (profiling runs)")
    ap.add_argument("--batched-only", action="store_true", help="profiling runs: no single-cell calls either, so every launch in a trace belongs to a batched step")
    ap.add_argument("--only-sparse", action="store_true", help="run only the sparse-model legs (sparse_section) and print their JSON")
    ap.add_argument("--only-c4", action="store_true", help="of the secondary legs run only configs[3] (C4: every rank predicts its share, one gather to rank 0)")
    ap.add_argument("--c4-cells", type=int, default=1250, help="cells per rank in the C4 leg (configs[3]: 10 000 cells over 8 GPUs)")
    ap.add_argument("--c4-points", type=int, default=N_TEST, help="test points per cell in the C4 leg")
    return ap.parse_args()


def worst_code(codes):
    """Exit code of a launch from its ranks' ``Popen.wait()`` values: a rank killed by signal S reports -S and counts as 128 + S
    (max([0, -6]) == 0 hid exactly the aborts this project needs to see)."""
    return max((128 - c) if c < 0 else c for c in codes)


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: this process -- which has made NO GPU call -- starts N fresh rank
    processes (RANK / LOCAL_RANK / WORLD_SIZE and a private rendezvous prefix in the environment), waits for them and exits
    with the worst of their codes -- a rank killed by a signal (Popen reports -SIGNUM: -6 / -11 after a GPU fault) counts as 128 + SIGNUM,
    never as "less than 0 = fine"; every rank's code goes to stderr.  Rank 0 inherits stdout (the JSON line).  No torch in any of them."""
    import shutil
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="gprx_bench_")
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), GPRX_ID_FILE=os.path.join(tmp, "rccl"),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        # wait for all of them; once one rank has ended badly the others get a minute to notice (they may be blocked in a collective
        # that rank will never join -- RCCL has no timeout) and are then ended, so that the launch always returns
        _time = time
        t_start, first_bad = _time.time(), None
        while any(pr.poll() is None for pr in procs):
            if first_bad is None and any(pr.poll() not in (None, 0) for pr in procs):
                first_bad = _time.time()
            if (first_bad is not None and _time.time() - first_bad > 60.0) or _time.time() - t_start > 3000.0:
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()  # (exactly the processes started above)
                break
            _time.sleep(0.2)
        codes = [pr.wait() for pr in procs]
        worst = worst_code(codes)
        if worst:
            for r, c in enumerate(codes):
                sys.stderr.write(f"bench.py: rank {r} ended with " + (f"signal {-c}" if c < 0 else f"exit code {c}") + "\n")
        return worst
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(tmp, ignore_errors=True)


def mapped_runtimes():
    """Distinct HIP / RCCL libraries mapped into this process (one HIP runtime per rank is the point of the torch-free launch)."""
    libs = set()
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                name = ln.split()[-1]
                if "libamdhip64" in name or "librccl" in name:
                    libs.add(name)
    except OSError:
        pass
    return sorted(libs)


def sparse_section(device, extra):
    """The sparse-model legs of the bench (rank 0, N = 1): fills extra["sparse_sgpr"] and the flat keys earlier rounds reported."""
    from gpras_amd.gpr import GPRAS
    from gpras_amd.synth import make_regression

    # ---- the sparse model: what gpras actually fits (SGPR behind gpr.py:299; hot loop gpr.py:147-173; sizes of the sweep in
    # production/analysis/cross_validation.py:100-126: modes 1..50, M 1..300) -- N = 4096, d = 10 ----
    # M <= 64 runs the five-launch evaluation of sgpr_fused.h and the Adam loop resident on the device; larger M the general launch
    # sequence with the host-stepped loop.  flops per evaluation of the loss alone: 2 M^2 N + 2 M^3 / 3.
    n_s, d_s = 4096, 10
    sp = {"shape": {"n_train": n_s, "d": d_s}, "by_M": {}}
    xs50, ys50, xt50 = make_regression(n_s, d_s, n_outputs=50, n_test=100000, config=6, unit=1)
    xs50, ys50 = xs50.astype(np.float64), ys50.astype(np.float64)

    def sgpr_rates(m_s, modes, reps):
        g_ = GPRAS("RBF", device=device)
        g_._init_models(xs50, ys50[:, :modes], m_s, "grid")
        u_ = np.arange(modes, dtype=np.int32)
        th_ = np.stack([mm.theta() for mm in g_.models])
        z_ = np.stack([mm.Z for mm in g_.models])
        for _ in range(3):  # (first evaluation eager, second captures the launch graph, third replays it)
            g_.engine.objective_batch(u_, th_, 15, True, zs=z_)
        t1_ = time.perf_counter()
        for _ in range(reps):
            g_.engine.objective_batch(u_, th_, 15, True, zs=z_)
        dt_ = (time.perf_counter() - t1_) / reps
        del g_
        return modes / dt_

    for m_s, launches in ((50, 5), (128, None), (300, None)):
        flops_eval = 2.0 * m_s * m_s * n_s + 2.0 * m_s ** 3 / 3.0
        row = {"loss_flops_per_evaluation": flops_eval,
               "launches_per_evaluation": launches if launches else "general launch sequence (M > 64): ~45 and growing with M / 64"}
        for modes in (1, 16, 50):
            r = sgpr_rates(m_s, modes, 40 if m_s == 50 else 6)
            row[f"loss_grad_evals_per_s_{modes}_modes"] = r
            row[f"loss_tflops_{modes}_modes"] = r * flops_eval / 1e12
        sp["by_M"][str(m_s)] = row
    extra["sgpr_n4096_d10_m50_loss_grad_evals_per_s"] = sp["by_M"]["50"]["loss_grad_evals_per_s_1_modes"]
    extra["sgpr_batched16_loss_grad_evals_per_s"] = sp["by_M"]["50"]["loss_grad_evals_per_s_16_modes"]
    # the reference's default fit: k-means Z, two-stage Adam 100 + 100 (gpr.py:112-127), all modes in lock step
    for modes in (1, 16, 50):
        best_fit = np.inf
        for _ in range(3):
            g_ = GPRAS("RBF", device=device)
            t1 = time.perf_counter()
            g_.fit(xs50, ys50[:, :modes], 50, "kmeans", "two-stage")
            best_fit = min(best_fit, time.perf_counter() - t1)
            evals_ = int(sum(mm.n_evals for mm in g_.models))
            if modes != 50:
                del g_
        sp[f"two_stage_fit_seconds_{modes}_modes_M50"] = best_fit
        sp[f"two_stage_fit_evaluations_{modes}_modes_M50"] = evals_
    extra["sgpr_n4096_d10_m50_two_stage_fit_seconds"] = sp["two_stage_fit_seconds_1_modes_M50"]
    extra["sgpr_16_modes_two_stage_fit_seconds_lockstep"] = sp["two_stage_fit_seconds_16_modes_M50"]
    extra["sgpr_units_per_s_lockstep"] = 16 / sp["two_stage_fit_seconds_16_modes_M50"]
    # sparse predict (gpr.py:336-339): 50 fitted modes at N* = 100 000 points through GPRAS.predict (host arrays in, host arrays out)
    g_.predict(xt50[:4096])
    tp_, tp_first = np.inf, None
    for _ in range(3):  # (the first call at this size also allocates the device staging: reported beside the best of three)
        t1 = time.perf_counter()
        pm_, pv_ = g_.predict(xt50)
        dt_ = time.perf_counter() - t1
        tp_first = dt_ if tp_first is None else tp_first
        tp_ = min(tp_, dt_)
    sp["predict_50_modes_M50_n_test_100000"] = {"seconds_host_to_host": tp_, "seconds_first_call": tp_first, "timing": "best of 3", "points_per_s_all_modes": 50 * xt50.shape[0] / tp_,
                                                 "finite": bool(np.all(np.isfinite(pm_)) and np.all(pv_ > 0))}
    del g_
    # a long Adam run, the cross-validation's regime (max_iter 5 000 - 10 000 there): 5 000 steps on all variables, 16 modes, M = 50;
    # the early stop (patience 50 at 1e-5 relative improvement) may end modes sooner -- evaluations are reported
    g_ = GPRAS("RBF", device=device)
    t1 = time.perf_counter()
    g_.fit(xs50, ys50[:, :16], 50, "kmeans", "adam", max_iter=5000)
    ta_ = time.perf_counter() - t1
    ev_ = [int(mm.n_evals) for mm in g_.models]
    sp["adam_5000_steps_16_modes_M50"] = {"seconds": ta_, "evaluations_per_mode_min_max": [min(ev_), max(ev_)], "evaluations_per_s": sum(ev_) / ta_,
                                          "microseconds_per_lockstep_step": 1e6 * ta_ / max(ev_)}
    del g_
    # more modes than 16 (the sweep goes to 50): from 17 cells on the resident loop runs two groups of cells on two streams, one launch apart
    # (gprx.hip sf_group_count); 1 000 steps each
    for modes in (28, 50):
        g_ = GPRAS("RBF", device=device)
        t1 = time.perf_counter()
        g_.fit(xs50, ys50[:, :modes], 50, "kmeans", "adam", max_iter=1000)
        ta_ = time.perf_counter() - t1
        ev_ = [int(mm.n_evals) for mm in g_.models]
        sp[f"adam_1000_steps_{modes}_modes_M50"] = {"seconds": ta_, "evaluations_per_mode_min_max": [min(ev_), max(ev_)], "evaluations_per_s": sum(ev_) / ta_,
                                                   "microseconds_per_lockstep_step": 1e6 * ta_ / max(ev_)}
        del g_
    # the same for M = 128 (host-stepped loop over the general launch sequence), 200 steps
    g_ = GPRAS("RBF", device=device)
    t1 = time.perf_counter()
    g_.fit(xs50, ys50[:, :16], 128, "kmeans", "adam", max_iter=200)
    ta_ = time.perf_counter() - t1
    ev_ = [int(mm.n_evals) for mm in g_.models]
    sp["adam_200_steps_16_modes_M128"] = {"seconds": ta_, "evaluations_per_s": sum(ev_) / ta_, "microseconds_per_lockstep_step": 1e6 * ta_ / max(ev_)}
    del g_
    extra["sparse_sgpr"] = sp
    # for comparison the older scheme, one engine + host thread per mode (workers=8)
    xs16, ys16 = xs50, ys50[:, :16]
    m_s = 50
    g8 = GPRAS("RBF", device=device)
    t1 = time.perf_counter()
    g8.fit(xs16, ys16, m_s, "kmeans", "two-stage", workers=8)
    t8 = time.perf_counter() - t1
    extra["sgpr_16_modes_two_stage_fit_seconds_workers8"] = t8
    extra["sgpr_units_per_s_workers8"] = 16 / t8
    extra["sgpr_loss_grad_evals_per_s_workers8"] = sum(m.n_evals for m in g8.models) / t8
    del g8


def main():
    args = parse_args()
    if args.only_sparse:  # development aid: the sparse legs alone, one JSON object
        from gpras_amd import _build

        _build.build()
        extra = {}
        sparse_section(0, extra)
        print(json.dumps(extra))
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    # RCCL and the HIP runtime print banners to stdout: keep stdout for the one JSON line only
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "RANK" in os.environ
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # Ranks are launched by `python -m torch.distributed.run` (the driver) or by spawn_ranks above; either way THIS process does not
    # import torch: RANK / LOCAL_RANK / WORLD_SIZE come from the environment, the 128-byte RCCL id travels through files
    # (gpras_amd.comm.file_rendezvous: every rank first reports that it can load RCCL, so nobody blocks in ncclCommInitRank
    # behind a rank that cannot), timing barriers and the maximum over ranks go through gprx_comm_* -- ONE HIP runtime and ONE
    # RCCL per rank (torch's own HIP / HSA / RCCL tree beside ROCm's was the source of the round-2 failures).
    # GPRX_BENCH_TORCH=1 selects the round-2 path instead: torch is imported FIRST (so that libgprx.so binds to the HIP runtime torch
    # loaded) and torch.distributed launches the collective's bootstrap.  There is no automatic fall-back from one to the other:
    # once libgprx has initialised ROCm's HIP runtime, torch finds no GPU in the same process (measured: "No HIP GPUs are
    # available") -- a failed torch-free rendezvous ends every rank with a message instead.
    torch = dist = comm = fx = None
    comm_error = ""
    launcher = "single process"
    use_torch = distributed and os.environ.get("GPRX_BENCH_TORCH") == "1"
    if use_torch:
        import torch  # noqa: PLC0415
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        launcher = "torch.distributed (GPRX_BENCH_TORCH=1)"

    import fcntl

    from gpras_amd import _build, _lib
    from gpras_amd._lib import DeviceBuffer, check, ptr
    from gpras_amd.synth import make_regression

    with open(os.path.join(ROOT, ".bench_build.lock"), "w") as lock:  # one rank builds (normally nothing to do), the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        _build.build()
    lib = _lib.load()
    device = local_rank if distributed else 0
    if distributed:
        from gpras_amd.comm import Communicator, default_id_prefix, report

        prefix = default_id_prefix()
        try:
            if os.environ.get("GPRX_BENCH_FORCE_COMM_FAILURE") == "1":  # (testing hook for the last-resort path below)
                raise RuntimeError("forced by GPRX_BENCH_FORCE_COMM_FAILURE")
            # (with a process group the id travels through it; without, through files)
            comm = Communicator.bootstrap(device, rank, world, id_file=None if use_torch else prefix)
            if not use_torch:
                launcher = "torch-free ranks: environment + file rendezvous, gprx_comm_* (RCCL behind the C ABI) for barriers, max and the gather"
        except Exception as exc:  # noqa: BLE001
            comm, comm_error = None, f"{type(exc).__name__}: {exc}"
        if use_torch:
            ok = torch.tensor([1 if comm is not None else 0], device=f"cuda:{local_rank}")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm, comm_error = None, comm_error or "another rank could not create its communicator"
        elif world > 1:
            said = report(prefix, "comm", rank, world, comm is not None)
            if None in said:  # a rank that is not there cannot take part in any exchange: end the run
                sys.stderr.write(f"bench.py rank {rank}: rank(s) {[r for r, v in enumerate(said) if v is None]} did not report -- giving up\n")
                sys.exit(3)
            if not all(said):
                if comm is not None:
                    comm.close()
                comm, comm_error = None, comm_error or "another rank could not create its communicator"
        if comm is None and not use_torch:
            # No communicator on every rank.  The line this run is for measures the path WITH its RCCL gather (north_star), so the default is
            # to end here, loudly: exit code 3 and every reason on stderr.  GPRX_BENCH_FILE_EXCHANGE=1 opts in to the shard-scaling
            # measurement without RCCL: the data path has no collective -- the ranks only meet at the timing barriers and for the final
            # gather of 8 bytes per cell -- and those meetings then go through files on this node; config.collective says so.
            sys.stderr.write(f"bench.py rank {rank}: the RCCL communicator could not be created on every rank ({comm_error})\n")
            if os.environ.get("GPRX_BENCH_FILE_EXCHANGE") != "1":
                sys.stderr.write(f"bench.py rank {rank}: giving up (exit 3).  GPRX_BENCH_FILE_EXCHANGE=1 measures the shards with barriers and the gather "
                                 "through files instead; GPRX_BENCH_TORCH=1 selects the torch.distributed launch path\n")
                sys.exit(3)
            from gpras_amd.comm import FileExchange

            fx = FileExchange(prefix, rank, world, timeout_s=180.0)  # (a rank that dies later ends the others after 3 minutes)
            launcher = "torch-free ranks: environment + file rendezvous; RCCL unavailable, so barriers, max and the gather through files (opt-in)"

    # ---- workload: `cells` independent cells per rank, seeds 1000 * config + unit ----
    # One handle per rank: x (N, d) and one y column per cell, resident in HBM before the timed region.
    from gpras_amd.model import NOISE_LOWER, softplus_inv

    cells = max(1, args.cells)
    x, y, xs = make_regression(N_TRAIN, DIM, n_outputs=cells, n_test=N_TEST, config=2, unit=rank)
    h = C.c_void_p()
    check(lib.gprx_create(device, N_TRAIN, DIM, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h)))
    check(lib.gprx_set_data(h, ptr(x), ptr(y), cells), h)
    # reference initial values: variance 1, lengthscale mean|x|, noise 1 (gpr.py:289, :298); the cells' hyperparameters
    # are spread around them (cell 0 keeps them exactly), so every cell builds and factors a different matrix
    theta = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
    spread = np.random.default_rng(2000 + rank).uniform(-0.15, 0.15, size=(cells, 3))
    spread[0] = 0.0
    thetas = np.ascontiguousarray(theta[None, :] + spread)
    units = np.arange(cells, dtype=np.int32)
    losses = np.zeros(cells)
    status = np.zeros(cells, dtype=np.int32)
    mask = 7
    loss = C.c_double()

    def sync_all():
        # device work of this rank done, then all ranks arrived (RCCL all-reduce + stream wait), then nothing left in flight
        check(lib.gprx_synchronize(h), h)
        if comm is not None:
            comm.barrier()
        elif fx is not None:
            fx.barrier()
        elif distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def fit_step():
        # one step: every cell of this rank fitted once by one batched launch sequence
        check(lib.gprx_factorize_batch(h, cells, ptr(units), ptr(thetas), mask, ptr(losses), ptr(status)), h)

    def fit_one():
        check(lib.gprx_factorize(h, 0, ptr(theta), None, mask, C.byref(loss)), h)

    d_mine = d_all = None
    if distributed:
        d_mine, d_all = DeviceBuffer(8 * cells, device), DeviceBuffer(8 * cells * world, device)
    for _ in range(args.warmup):
        fit_step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit_step()
    if distributed:
        # the single collective of the job: every rank's results gathered over RCCL (gprx_comm_all_gather, device buffers)
        if comm is not None:
            check(lib.gprx_memcpy_h2d(device, d_mine.ptr, ptr(losses), losses.nbytes))
            comm.all_gather_dev(d_mine, d_all, cells)
            comm.synchronize()
        elif fx is not None:
            fx_parts = fx.all_gather(losses)
        else:
            mine = torch.tensor(losses, dtype=torch.float64, device=f"cuda:{local_rank}")
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        all_losses = d_all.to_array((world, cells)) if comm is not None else np.stack(fx_parts) if fx is not None else torch.stack(gathered).cpu().numpy()
        assert np.array_equal(all_losses[rank], losses) and np.all(np.isfinite(all_losses))
        if comm is not None:
            elapsed = comm.max(elapsed)  # the slowest rank's time
        elif fx is not None:
            elapsed = fx.max(elapsed)
        else:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
    fits_per_s = world * cells * args.steps / elapsed
    # "did RCCL see N ranks" from the record itself: ncclCommCount of the communicator, and the distinct rank numbers that came back
    # through an RCCL all-gather of every rank's ncclCommUserRank
    rccl_ranks_seen = None
    if comm is not None:
        r_seen, w_seen = comm.rank_and_world_seen_by_rccl()
        rccl_ranks_seen = {"ncclCommCount": w_seen, "distinct_ranks_gathered": len({int(v[0]) for v in comm.all_gather(np.array([float(r_seen)]))})}

    result = {
        "metric": "gp_fits_per_sec",
        "value": fits_per_s,
        "unit": "fits/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] as the many-independent-cells workload: exact GP N=4096 d=8 RBF, fit F1 = kernel build + blocked fp64-MFMA Cholesky + alpha + LML, per cell; cells = y columns of one training set with per-cell hyperparameters (the reference's per-mode models)",
            "n_train": N_TRAIN,
            "d": DIM,
            "kernel": "RBF",
            "cells_per_gpu_per_step": cells,
            "parallelism": f"{cells} independent cells per batched launch sequence per GPU x {world} GPU, one RCCL all_gather at the end",
            "collective": ("none (one process)" if not distributed else "gprx_comm_all_gather (RCCL behind the C ABI, device-resident buffers)" if comm is not None
                           else f"FILE EXCHANGE on this node, not RCCL (the communicator could not be created: {comm_error})" if fx is not None
                           else f"torch.distributed all_gather (fallback: {comm_error})"),
            "launcher": launcher if (comm is not None or fx is not None or not distributed) else f"{launcher}; gather through torch.distributed ({comm_error})",
            "rccl_ranks_seen": rccl_ranks_seen,
            "hip_and_rccl_libraries_mapped": mapped_runtimes(),
        },
    }

    # ---- BASELINE configs[3] (C4), MEASURED at one GPU's full share: `--c4-cells` (1250 = 10 000 / 8) independent cells, each fitted and
    # predicted (mean + variance, noise included) at the shared `--c4-points` (100 000) test points through gprx_predict_batch_dev into ONE
    # device block [means (cells, N*) | variances (cells, N*)] (2.0 GB), in chunks that fit the free HBM next to the factors (L and L^-1 of a
    # chunk's cells: gprx_cell_bytes); then the ONE collective north_star names: gprx_comm_gather of every rank's block to rank 0 over RCCL
    # (reference loop: gpr.py:336-341).  Every rank runs it (so that a launch with N ranks measures N shares and the real gather).
    c4 = None
    if (not args.no_extras and not args.batched_only) or args.only_c4:
        try:
            n_c4, ns4 = max(1, args.c4_cells), max(1, min(args.c4_points, N_TEST))
            xs4 = np.ascontiguousarray(xs[:ns4])
            dxs4 = DeviceBuffer.from_array(xs4, device)
            block_elems = 2 * n_c4 * ns4
            block = DeviceBuffer(8 * block_elems, device)
            nb, fr, tot = C.c_int64(), C.c_int64(), C.c_int64()
            check(lib.gprx_cell_bytes(h, 1, C.byref(nb)), h)
            check(lib.gprx_mem_info(device, C.byref(fr), C.byref(tot)))
            # (the arena of `cells` slots exists already: a chunk of at most `cells` cells only adds the L^-1 workspace, which gprx_cell_bytes
            # over-counts by the arena cell -- safe side)
            chunk = int(max(1, min(cells, n_c4, (fr.value - 0.1 * tot.value) // max(nb.value, 1))))
            units4 = np.ascontiguousarray(np.arange(n_c4, dtype=np.int32) % cells)
            spread4 = np.random.default_rng(3000 + rank).uniform(-0.15, 0.15, size=(n_c4, 3))
            spread4[0] = 0.0
            thetas4 = np.ascontiguousarray(theta[None, :] + spread4)
            check(lib.gprx_predict_batch_dev(h, 1, ptr(units4), ptr(thetas4), None, dxs4.ptr, min(ns4, 1024), block.ptr, block.at(n_c4 * ns4), 1), h)  # (allocations)
            sync_all()
            t1 = time.perf_counter()
            for lo in range(0, n_c4, chunk):
                cnt = min(chunk, n_c4 - lo)
                u_part, t_part = units4[lo:lo + cnt], thetas4[lo:lo + cnt]
                check(lib.gprx_predict_batch_dev(h, cnt, ptr(u_part), ptr(t_part), None, dxs4.ptr, ns4, block.at(lo * ns4), block.at((n_c4 + lo) * ns4), 1), h)
                if rank == 0:  # (gprx_predict_batch_dev returns once the chunk is ENQUEUED: these are not completion times)
                    sys.stderr.write(f"bench.py: C4 leg: chunk up to cell {lo + cnt} / {n_c4} enqueued\n")
            check(lib.gprx_synchronize(h), h)
            t_c4 = time.perf_counter() - t1
            if rank == 0:
                sys.stderr.write(f"bench.py: C4 leg finished: {n_c4} cells x {ns4} points in {t_c4:.1f} s (after the final synchronize)\n")
            flops4 = n_c4 * (float(N_TRAIN) ** 2 * ns4 + 2.0 * N_TRAIN * ns4 + 2.0 * N_TRAIN ** 3 / 3)  # predict + (Cholesky + L^-1) per cell
            c4 = {"cells_per_gpu": n_c4, "n_test": ns4, "cells_per_chunk": chunk, "block_GB_per_gpu": 8e-9 * block_elems,
                  "seconds_this_gpu": t_c4, "cells_per_s_per_gpu": n_c4 / t_c4, "points_per_s_per_gpu": n_c4 * ns4 / t_c4,
                  "tflops_predict_plus_factor_plus_inverse": flops4 / t_c4 / 1e12}
            c4["frac_of_fp64_mfma_peak"] = c4["tflops_predict_plus_factor_plus_inverse"] / FP64_MFMA_PEAK_TFLOPS
            c4_cell0 = (block.to_array((min(ns4, 2000),)), np.empty(min(ns4, 2000)))  # cell 0 (the reference's initial hyperparameters): mean, variance
            check(lib.gprx_memcpy_d2h(device, ptr(c4_cell0[1]), block.at(n_c4 * ns4), c4_cell0[1].nbytes))
            if comm is not None:
                t_all = comm.max(t_c4)  # the slowest rank's share
                recv = DeviceBuffer(8 * block_elems * world, device) if rank == 0 else None
                comm.barrier()
                t1 = time.perf_counter()
                comm.gather_dev(block, recv, block_elems, 0)
                comm.synchronize()
                t_g = time.perf_counter() - t1
                comm.barrier()
                c4["seconds_slowest_gpu"] = t_all
                c4["cells_per_s_all_gpus"] = world * n_c4 / t_all
                c4["gather_seconds_on_root"] = t_g
                inbound = 8.0 * block_elems * max(world - 1, 1)
                c4["C4_gather_GBps"] = inbound / t_g / 1e9
                c4["gather_note"] = (f"gprx_comm_gather (grouped ncclSend / ncclRecv) of {world} blocks of {8e-9 * block_elems:.2f} GB to rank 0; GB/s = bytes arriving at the root from the "
                                     f"other {world - 1} rank(s) / time on the root" if world > 1 else
                                     "world of one: gprx_comm_gather degenerates to the root's own send/receive pair (a device copy), no xGMI traffic")
                if rank == 0:
                    ok_rows = True
                    for r in range(world):  # cell 0 of every rank's block arrived (own block: equal to what was sent)
                        got = np.empty(min(ns4, 2000))
                        check(lib.gprx_memcpy_d2h(device, ptr(got), recv.at(r * block_elems), got.nbytes))
                        ok_rows = ok_rows and bool(np.all(np.isfinite(got))) and (r != 0 or np.array_equal(got, c4_cell0[0]))
                    c4["gathered_blocks_checked"] = ok_rows
                    recv.free()
            else:
                c4["seconds_slowest_gpu"] = fx.max(t_c4) if fx is not None else t_c4
                c4["C4_gather_GBps"] = None
                c4["gather_note"] = "no RCCL communicator in this run: the gather leg was not measured"
            c4["seconds_for_10k_cells_on_8_gpus"] = (c4["seconds_slowest_gpu"] * (1250.0 / n_c4) * (100000.0 / ns4)) if (n_c4, ns4) != (1250, 100000) else c4["seconds_slowest_gpu"]
            c4["seconds_for_10k_cells_on_8_gpus_is"] = ("measured: this rank count x 1250 cells x 100 000 points" if (n_c4, ns4) == (1250, 100000) else "SCALED from a reduced --c4-cells / --c4-points run") + ("" if world == 8 else f"; with {world} of the 8 GPUs present, each running one GPU's full share (shards are independent)")
            block.free()
            dxs4.free()
        except Exception as exc:  # noqa: BLE001  (a rank that fails here cannot take part in the gather: the others must not wait for it)
            import traceback

            traceback.print_exc()
            if world > 1:
                sys.stderr.write(f"bench.py rank {rank}: the C4 leg failed ({type(exc).__name__}: {exc}); ending the run so that no rank waits in the gather\n")
                sys.stderr.flush()
                os._exit(4)
            c4 = {"error": f"{type(exc).__name__}: {exc}"}
        if c4 is not None:
            result["C4"] = c4

    if rank == 0:
        # ---- roofline of the dominant kernel: instrumented pass over the same batched step, HIP events around every launch ----
        check(lib.gprx_set_profiling(h, 1), h)
        prof = (C.c_double * 8)()
        acc = np.zeros(8)
        reps = 3
        kb_ms, kb_bytes = C.c_double(), C.c_double()
        kb_acc = 0.0
        for _ in range(reps):
            fit_step()
            lib.gprx_last_profile(h, prof)
            acc += np.array(list(prof))
            lib.gprx_last_kernel_build(h, C.byref(kb_ms), C.byref(kb_bytes))
            kb_acc += kb_ms.value
        check(lib.gprx_set_profiling(h, 0), h)
        kb_ms_avg = kb_acc / reps
        # the kernel build of the judged workload: ONE launch writes the lower tiles of all cells (HIP events around that launch)
        result["kernel_build_hbm"] = {
            "kernel": f"gprx::kmat_kernel<0,0>: K(X, X) + noise on the diagonal, the 64 x 64 tiles on or below the diagonal of {cells} cells in one launch",
            "bound": "hbm", "unit": "GB/s", "bytes_written_per_launch": kb_bytes.value, "avg_launch_us": 1e3 * kb_ms_avg,
            "GBps": kb_bytes.value / (kb_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "frac_of_8TBps": kb_bytes.value / (kb_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        gemm_ms, gemm_launches, gemm_flops, panel_ms, panel_launches, strip_ms, strip_launches, strip_flops = acc / reps
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
        result["roofline"] = {
            "kernel": f"gprx::gemm_f64_kernel<0,1,64,64,0,0,1> (NT, 64 x 64 tile, both operands by LDS-DMA): every launch of the Cholesky's main update kernel A22 -= L21 L21^T in a step, all {cells} cells per launch (5 bulk HEAD/TAIL updates with K = 1024 + 12 in-block updates with K = 256 / 512)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": pmc_traffic(MAIN_KERNEL),
            "traffic_source": f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command; 2 x FETCH_SIZE + WRITE_SIZE per launch), not measured in this run",
            "launches_per_step": gemm_launches,
            "avg_launch_us": 1e3 * gemm_ms / gemm_launches,
            "algorithmic_flops_per_launch": gemm_flops / gemm_launches,
            "algorithmic_flops_per_step": gemm_flops,
            "panel_kernel": {"launches_per_step": panel_launches, "avg_launch_us": 1e3 * panel_ms / panel_launches},
            "short_k_inblock_updates": {"kernels": "gemm_f64_kernel<0,1,64,64,1,0,1> (K = 64 and K = 128: LDS-DMA operands, C prefetched)", "launches_per_step": strip_launches, "avg_launch_us": 1e3 * strip_ms / max(strip_launches, 1), "tflops": strip_flops / (strip_ms * 1e-3) / 1e12 if strip_ms else None},
            "cholesky_flops_per_step": cells * N_TRAIN**3 / 3,
            "whole_step_tflops": cells * N_TRAIN**3 / 3 / (elapsed / args.steps) / 1e12,
        }
    if rank == 0 and not args.batched_only:
        fit_one()
        ms = (C.c_double * 4)()
        lib.gprx_last_timings(h, ms)
        kmat_bytes = 8.0 * (N_TRAIN * (N_TRAIN + 64) / 2) + 8.0 * N_TRAIN * DIM
        result["stages_ms"] = {"kernel_build": ms[0], "cholesky": ms[1], "solves": ms[2]}
        # single-cell latency (one cell alone on the GPU, look-ahead on two streams)
        fit_one()
        t1 = time.perf_counter()
        for _ in range(10):
            fit_one()
        result["single_cell_ms_per_fit"] = 1e3 * (time.perf_counter() - t1) / 10
        # one cell alone: the kernel-build launch by itself (events around it) and the whole stage (lengthscale upload included)
        check(lib.gprx_set_profiling(h, 1), h)
        fit_one()
        k1_ms, k1_bytes = C.c_double(), C.c_double()
        lib.gprx_last_kernel_build(h, C.byref(k1_ms), C.byref(k1_bytes))
        check(lib.gprx_set_profiling(h, 0), h)
        result.setdefault("kernel_build_hbm", {})["single_cell"] = {
            "launch_us": 1e3 * k1_ms.value, "GBps": k1_bytes.value / (k1_ms.value * 1e-3) / 1e9 if k1_ms.value > 0 else None,
            "frac_of_8TBps": k1_bytes.value / (k1_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS if k1_ms.value > 0 else None,
            "stage_ms_with_parameter_upload": ms[0], "stage_GBps": kmat_bytes / (ms[0] * 1e-3) / 1e9}

    x1 = y1 = x5 = y5 = xs5 = gpu_mean1 = gpu_mean5 = None
    if rank == 0 and not args.no_extras and not args.batched_only and not args.only_c4:
        # (the secondary measurements must never cost the headline line: a failure is reported, not raised)
        try:
            extra = {}
            # F2: objective + gradient
            grad = np.zeros(3)
            check(lib.gprx_objective(h, 0, ptr(theta), None, mask, C.byref(loss), ptr(grad)), h)
            t1 = time.perf_counter()
            k2 = max(3, args.steps // 4)
            for _ in range(k2):
                check(lib.gprx_objective(h, 0, ptr(theta), None, mask, C.byref(loss), ptr(grad)), h)
            extra["F2_objective_grad_evals_per_s"] = k2 / (time.perf_counter() - t1)
            # F2 batched: the same evaluation for all cells of the step by batched launches (gprx_objective_batch)
            gl, gg = np.zeros(cells), np.zeros((cells, 3))
            check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), None, mask, ptr(gl), ptr(gg)), h)
            t1 = time.perf_counter()
            for _ in range(3):
                check(lib.gprx_objective_batch(h, cells, ptr(units), ptr(thetas), None, mask, ptr(gl), ptr(gg)), h)
            extra["F2_batched_objective_grad_evals_per_s"] = 3 * cells / (time.perf_counter() - t1)
            extra["F2_batched_tflops"] = extra["F2_batched_objective_grad_evals_per_s"] * N_TRAIN**3 / 1e12
            # predict: mean + variance at 100k points, inputs and outputs resident in HBM
            fit_one()
            dxs = DeviceBuffer.from_array(xs, device)
            dmean, dvar = DeviceBuffer(8 * N_TEST, device), DeviceBuffer(8 * N_TEST, device)
            def predict_rate(handle, dx, ntest, dm, dv, n_train):
                """device-resident predict (mean + variance, noise included) at `ntest` points of the handle's current factorisation: best of 3
                after a warm call (which also forms L^-1); flops = N^2 N* + 2 N N* """
                best_t = np.inf
                for rep in range(4):
                    t1 = time.perf_counter()
                    check(lib.gprx_predict_dev(handle, dx.ptr, ntest, dm.ptr, dv.ptr, 1), handle)
                    check(lib.gprx_synchronize(handle), handle)
                    if rep:
                        best_t = min(best_t, time.perf_counter() - t1)
                tf = (float(n_train) ** 2 * ntest + 2.0 * n_train * ntest) / best_t / 1e12
                return {"points_per_s": ntest / best_t, "tflops": tf, "frac_of_fp64_mfma_peak": tf / FP64_MFMA_PEAK_TFLOPS, "n_test": ntest, "timing": "best of 3"}

            pr4 = predict_rate(h, dxs, N_TEST, dmean, dvar, N_TRAIN)
            extra["predict_points_per_s"] = pr4["points_per_s"]
            extra["predict_tflops"] = pr4["tflops"]
            extra["predict_frac_of_fp64_mfma_peak"] = pr4["frac_of_fp64_mfma_peak"]
            extra["predict_timing"] = "best of 3, device-resident"
            extra["predict_n_test"] = N_TEST
            gpu_mean = dmean.to_array((N_TEST,))[:2000]
            gpu_var = dvar.to_array((N_TEST,))[:2000]
            # F3: BASELINE configs[2] -- Matern-5/2 ARD, 50 L-BFGS-B iterations on the exact LML
            from gpras_amd.gpr import GPRAS

            g3 = GPRAS("Matern52", device=device)
            t1 = time.perf_counter()
            g3.fit(x, y[:, :1], None, optimization_method="L-BFGS-B", ard=True, max_iter=50)
            t3 = time.perf_counter() - t1
            extra["F3_lbfgs50_matern52_ard_seconds"] = t3
            extra["F3_evaluations"] = g3.models[0].n_evals
            extra["F3_fits_per_s"] = 1.0 / t3
            # F3 over 16 modes of one training set in lock step (batched evaluations; bit-identical to the serial loop)
            g16 = GPRAS("Matern52", device=device)
            t1 = time.perf_counter()
            g16.fit(x, y[:, :16], None, optimization_method="L-BFGS-B", ard=True, max_iter=50)
            t16 = time.perf_counter() - t1
            extra["F3_lockstep_16_modes_seconds"] = t16
            extra["F3_lockstep_fits_per_s"] = 16 / t16
            extra["F3_lockstep_evaluations"] = int(sum(m.n_evals for m in g16.models))
            del g16
            sparse_section(device, extra)
            # the other sizes of the target: N = 1024 (batched cells) and BASELINE configs[4], N = 16384 d = 12 (one cell alone)
            sizes = {}
            c1 = 512  # smaller matrices need more cells per launch to fill the chip
            x1, y1, _ = make_regression(1024, DIM, n_outputs=c1, n_test=0, config=2, unit=500)
            h1 = C.c_void_p()
            check(lib.gprx_create(device, 1024, DIM, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h1)))
            check(lib.gprx_set_data(h1, ptr(x1), ptr(y1), c1), h1)
            units1 = np.arange(c1, dtype=np.int32)
            thetas1 = np.ascontiguousarray(np.tile(thetas, (c1 // cells + 1, 1))[:c1])
            losses1, status1 = np.zeros(c1), np.zeros(c1, dtype=np.int32)
            for _ in range(2):
                check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1)
            t1 = time.perf_counter()
            for _ in range(10):
                check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1)
            sizes["N1024_d8_batched_fits_per_s"] = 10 * c1 / (time.perf_counter() - t1)
            sizes["N1024_d8_cells_per_launch"] = c1
            sizes["N1024_d8_batched_tflops"] = sizes["N1024_d8_batched_fits_per_s"] * 1024**3 / 3 / 1e12
            sizes["N1024_d8_batched_path"] = "one workgroup per cell (potrf_cell.h: the default from 256 cells of N <= 1024); the launch sequence beside it below"
            check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", -1), h1)
            for _ in range(2):
                check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1)
            t1 = time.perf_counter()
            for _ in range(10):
                check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1)
            sizes["N1024_d8_batched_fits_per_s_launch_sequence"] = 10 * c1 / (time.perf_counter() - t1)
            check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", 0), h1)

            def roofline_block(handle, run, label):
                """the main update kernel's launches of one instrumented pass (HIP events around every launch, as the headline block)"""
                check(lib.gprx_set_profiling(handle, 1), handle)
                run()
                pr = (C.c_double * 8)()
                lib.gprx_last_profile(handle, pr)
                check(lib.gprx_set_profiling(handle, 0), handle)
                g_ms, g_n, g_fl = pr[0], pr[1], pr[2]
                if g_n <= 0 or g_ms <= 0:
                    return {"kernel": label, "note": "no launch of the main update kernel at this size"}
                ach = g_fl / (g_ms * 1e-3) / 1e12
                return {"kernel": label, "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                        "launches": g_n, "avg_launch_us": 1e3 * g_ms / g_n, "algorithmic_flops_per_launch": g_fl / g_n,
                        "panel_launches": pr[4], "panel_avg_launch_us": 1e3 * pr[3] / max(pr[4], 1.0),
                        "short_k_launches": pr[6], "short_k_tflops": (pr[7] / (pr[5] * 1e-3) / 1e12) if pr[5] > 0 else None}

            # the kernel that IS the default at this size: ONE launch of the one-workgroup-per-cell Cholesky (potrf_cell2_kernel), timed by HIP
            # events around that launch on its stream (gprx_last_cell_kernel; the handle forces the kernel so that the profiled pass runs it)
            check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", 1), h1)
            check(lib.gprx_set_profiling(h1, 1), h1)
            cms, cfl, ccells, best_c = C.c_double(), C.c_double(), C.c_double(), None
            try:
                for _ in range(3):
                    check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1)
                    check(lib.gprx_last_cell_kernel(h1, C.byref(cms), C.byref(cfl), C.byref(ccells)), h1)
                    if cms.value > 0 and (best_c is None or cms.value < best_c[0]):
                        best_c = (cms.value, cfl.value, ccells.value)
            finally:
                check(lib.gprx_set_profiling(h1, 0), h1)
                check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", 0), h1)
            if best_c:
                ach = best_c[1] / (best_c[0] * 1e-3) / 1e12
                # HBM bytes per launch from the committed PMC passes over this kernel (separate FETCH_SIZE / WRITE_SIZE runs, 2 x FETCH + WRITE)
                traffic = None
                try:
                    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_pmc_cell_kernel.json")) as f:
                        traffic = float(next(v for k, v in json.load(f)["kernels"].items() if k.startswith("potrf_cell2_kernel"))["hbm_bytes_per_launch_corrected"])
                except Exception:  # noqa: BLE001
                    pass
                sizes["N1024_d8_roofline"] = {
                    "kernel": "gprx::potrf_cell2_kernel<false,true>: the whole Cholesky of 512 cells of N = 1024 and the forward substitution of their right-hand sides in ONE launch, one workgroup per cell (column pairs, LDS-DMA operand panels); best of 3",
                    "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                    "launch_us": 1e3 * best_c[0], "cells": best_c[2], "algorithmic_flops_per_launch": best_c[1],
                    "traffic": traffic, "traffic_source": "profiles/r04_pmc_cell_kernel.json (rocprofv3 --pmc passes of tools/pmc_cell.sh), not measured in this run",
                    "hbm_GBps_at_that_traffic": (traffic / (best_c[0] * 1e-3) / 1e9) if traffic else None}
            check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", -1), h1)
            sizes["N1024_d8_roofline_launch_sequence"] = roofline_block(
                h1, lambda: check(lib.gprx_factorize_batch(h1, c1, ptr(units1), ptr(thetas1), mask, ptr(losses1), ptr(status1)), h1),
                "gemm_f64_kernel<0,1,64,64,0,0,1>: the K = 256 / 512 in-block updates of the LAUNCH SEQUENCE (not the default at this size), 512 cells per launch (the whole matrix is one outer block)")
            check(lib.gprx_set_handle_tuning(h1, b"cell_kernel", 0), h1)
            check(lib.gprx_factorize(h1, 0, ptr(theta), None, mask, C.byref(loss)), h1)
            t1 = time.perf_counter()
            for _ in range(10):
                check(lib.gprx_factorize(h1, 0, ptr(theta), None, mask, C.byref(loss)), h1)
            sizes["N1024_d8_single_cell_ms"] = 1e2 * (time.perf_counter() - t1)
            # predictive mean + variance at the other sizes of the target (north_star: N in {1k, 4k, 16k}), same 100 000 points
            pr1 = predict_rate(h1, dxs, N_TEST, dmean, dvar, 1024)
            sizes["N1024_d8_predict_points_per_s"] = pr1["points_per_s"]
            sizes["N1024_d8_predict"] = pr1
            gpu_mean1, gpu_var1 = dmean.to_array((2000,)), dvar.to_array((2000,))
            lib.gprx_destroy(h1)
            n5, d5 = 16384, 12
            x5, y5, xs5 = make_regression(n5, d5, n_outputs=1, n_test=N_TEST, config=5, unit=0)
            h5 = C.c_void_p()
            check(lib.gprx_create(device, n5, d5, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h5)))
            check(lib.gprx_set_data(h5, ptr(x5), ptr(y5), 1), h5)
            th5 = np.ascontiguousarray([softplus_inv(1.0), softplus_inv(np.mean(np.abs(x5))), softplus_inv(1.0 - NOISE_LOWER)], dtype=np.float64)
            check(lib.gprx_factorize(h5, 0, ptr(th5), None, mask, C.byref(loss)), h5)
            t1 = time.perf_counter()
            for _ in range(3):
                check(lib.gprx_factorize(h5, 0, ptr(th5), None, mask, C.byref(loss)), h5)
            t5 = (time.perf_counter() - t1) / 3
            ms5 = (C.c_double * 4)()
            lib.gprx_last_timings(h5, ms5)
            sizes["N16384_d12_fit_ms"] = 1e3 * t5
            sizes["N16384_d12_fits_per_s"] = 1.0 / t5
            sizes["N16384_d12_cholesky_tflops"] = n5**3 / 3 / (ms5[1] * 1e-3) / 1e12
            sizes["N16384_d12_cholesky_frac_of_fp64_mfma_peak"] = sizes["N16384_d12_cholesky_tflops"] / FP64_MFMA_PEAK_TFLOPS
            sizes["N16384_d12_kernel_build_GBps"] = (8.0 * n5 * (n5 + 64) / 2 + 8.0 * n5 * d5) / (ms5[0] * 1e-3) / 1e9
            sizes["N16384_d12_roofline"] = roofline_block(
                h5, lambda: check(lib.gprx_factorize(h5, 0, ptr(th5), None, mask, C.byref(loss)), h5),
                "gemm_f64_kernel<0,1,64,64,0,0,1>: bulk HEAD / TAIL updates with K = 512 and the in-block updates with K >= 256 of ONE matrix")
            dxs5 = DeviceBuffer.from_array(xs5, device)
            pr5 = predict_rate(h5, dxs5, N_TEST, dmean, dvar, n5)
            sizes["N16384_d12_predict_points_per_s"] = pr5["points_per_s"]
            sizes["N16384_d12_predict"] = pr5
            gpu_mean5, gpu_var5 = dmean.to_array((2000,)), dvar.to_array((2000,))
            dxs5.free()
            lib.gprx_destroy(h5)
            # small matrices in many cells: one workgroup per cell (potrf_cell.h; defaults by size: gprx.hip use_cell_kernel)
            c6 = 512
            x6, y6, _ = make_regression(512, DIM, n_outputs=c6, n_test=0, config=2, unit=600)
            units6 = np.arange(c6, dtype=np.int32)
            thetas6 = np.ascontiguousarray(np.tile(thetas, (c6 // cells + 1, 1))[:c6])
            losses6, status6 = np.zeros(c6), np.zeros(c6, dtype=np.int32)
            for key, knob in (("launch_sequence", -1), ("one_workgroup_per_cell", 1)):
                h6 = C.c_void_p()
                check(lib.gprx_create(device, 512, DIM, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h6)))
                check(lib.gprx_set_handle_tuning(h6, b"cell_kernel", knob), h6)  # (this handle only: no process-wide knob to restore)
                check(lib.gprx_set_data(h6, ptr(x6), ptr(y6), c6), h6)
                for _ in range(2):
                    check(lib.gprx_factorize_batch(h6, c6, ptr(units6), ptr(thetas6), mask, ptr(losses6), ptr(status6)), h6)
                t1 = time.perf_counter()
                for _ in range(10):
                    check(lib.gprx_factorize_batch(h6, c6, ptr(units6), ptr(thetas6), mask, ptr(losses6), ptr(status6)), h6)
                sizes[f"N512_d8_batched_fits_per_s_{key}"] = 10 * c6 / (time.perf_counter() - t1)
                lib.gprx_destroy(h6)
            # the opt-in tile-DAG factorisation of a lone matrix (potrf_dag.h), for the record beside single_cell_ms_per_fit
            h7 = C.c_void_p()
            check(lib.gprx_create(device, N_TRAIN, DIM, 0, _lib.KERNEL_IDS["RBF"], 0, C.byref(h7)))
            check(lib.gprx_set_data(h7, ptr(x), ptr(y[:, :1].copy()), 1), h7)
            check(lib.gprx_set_handle_tuning(h7, b"dag", 1), h7)
            for _ in range(3):
                check(lib.gprx_factorize(h7, 0, ptr(theta), None, mask, C.byref(loss)), h7)
            t1 = time.perf_counter()
            for _ in range(10):
                check(lib.gprx_factorize(h7, 0, ptr(theta), None, mask, C.byref(loss)), h7)
            sizes["N4096_single_cell_ms_per_fit_tile_dag_opt_in"] = 1e2 * (time.perf_counter() - t1)
            lib.gprx_destroy(h7)
            # k-means inducing-point initialisation (seeding + Lloyd iterations on the device), N = 16384, d = 10, M = 50
            from gpras_amd.kmeans import kmeans_centers

            xk, _, _ = make_regression(16384, 10, n_outputs=1, n_test=0, config=8, unit=16384)
            kmeans_centers(xk, 50, device=device)
            t1 = time.perf_counter()
            kmeans_centers(xk, 50, device=device)
            sizes["kmeans_init_N16384_d10_M50_ms"] = 1e3 * (time.perf_counter() - t1)
            extra["other_sizes"] = sizes
            # N1: EOF projection either side of the GP path, device-resident: transform (T, cells) -> (T, k)
            # and reverse (T, k) -> mean + variance fields (T, cells); HBM-bound, rates against the algorithmic bytes
            from gpras_amd.preprocess import EOFProjector
            from gpras_amd.synth import make_eof_state
            from oracle import pca as opca

            t_rows, n_cells, k_modes = 512, 200_000, 10
            st = make_eof_state(n_cells, k_modes, 64, seed=7)
            big = np.tile(st["x"], (t_rows // 64, 1))
            proj = EOFProjector(st["dry"], st["elevations"], st["input_mean"], st["weights"], st["eofs"], st["x_mean"], st["x_std"], "wse", device=device)
            cp = (n_cells + 15) // 16 * 16
            padded = np.zeros((t_rows, cp))
            padded[:, :n_cells] = big
            dx = DeviceBuffer.from_array(padded, device)
            dz = DeviceBuffer(8 * t_rows * k_modes, device)
            dfull, dvfull = DeviceBuffer(8 * t_rows * n_cells, device), DeviceBuffer(8 * t_rows * n_cells, device)
            times = {"transform": [], "reverse": []}
            for rep in range(4):
                t1 = time.perf_counter()
                check(lib.gprx_pca_transform_dev(proj.handle, dx.ptr, t_rows, dz.ptr))
                check(lib.gprx_pca_synchronize(proj.handle))
                times["transform"].append(time.perf_counter() - t1)
                t1 = time.perf_counter()
                check(lib.gprx_pca_reverse_dev(proj.handle, dz.ptr, dz.ptr, t_rows, dfull.ptr, dvfull.ptr))
                check(lib.gprx_pca_synchronize(proj.handle))
                times["reverse"].append(time.perf_counter() - t1)
            tt, tr = min(times["transform"][1:]), min(times["reverse"][1:])
            zs = dz.to_array((t_rows, k_modes))[:64]
            args = (st["dry"], st["elevations"], st["input_mean"], st["weights"], st["eofs"], st["x_mean"], st["x_std"], "wse")
            t1 = time.perf_counter()
            zr = opca.transform(st["x"], *args)
            tc = time.perf_counter() - t1
            t1 = time.perf_counter()
            opca.reverse_transform(zr, np.abs(zr), *args)
            tcr = time.perf_counter() - t1
            extra["eof_projection"] = {
                "shape": {"rows": t_rows, "cells": n_cells, "modes": k_modes},
                "transform_ms": 1e3 * tt,
                "transform_GBps_of_input_read_once": 8.0 * t_rows * n_cells / tt / 1e9,
                "reverse_mean_var_ms": 1e3 * tr,
                "reverse_GBps_of_output_written_once": 16.0 * t_rows * n_cells / tr / 1e9,
                "parity_transform_rel_err_vs_oracle": float(np.max(np.abs(zs - zr)) / np.max(np.abs(zr))),
                "cpu_oracle_rows_per_s": {"transform": 64 / tc, "reverse": 64 / tcr},
                "gpu_rows_per_s": {"transform": t_rows / tt, "reverse": t_rows / tr},
            }
            # N3: fused metrics over the reconstructed fields (truth / prediction / confidence: three (T, cells) fields, read once)
            drow, dcell, darg = DeviceBuffer(8 * t_rows * 4, device), DeviceBuffer(8 * 5 * n_cells, device), DeviceBuffer(4 * 2 * n_cells, device)
            nmatch = C.c_uint64()
            tmet = []
            for rep in range(4):
                t1 = time.perf_counter()
                check(lib.gprx_metrics_dev(device, dfull.ptr, dx.ptr, dvfull.ptr, t_rows, n_cells, 2, 0.05, drow.ptr, dcell.ptr, darg.ptr, C.byref(nmatch)))
                tmet.append(time.perf_counter() - t1)
            tm = min(tmet[1:])
            extra["field_metrics"] = {
                "shape": {"rows": t_rows, "cells": n_cells, "t_tol": 2},
                "ms": 1e3 * tm,
                "GBps_of_three_fields_read_once": 24.0 * t_rows * n_cells / tm / 1e9,
            }
            proj.close()
            for b in (dx, dz, dfull, dvfull, drow, dcell, darg):
                b.free()
            # the tail of production/analysis/pipeline.py (:256-288) resident in HBM (gpras_amd.pipeline) against the same steps through
            # host arrays: 10 sparse modes, 500 test timesteps, 50 000 cells
            from gpras_amd.pipeline import DevicePipeline
            from gpras_amd.preprocess import EOFProjector as _Proj

            prng = np.random.default_rng(77)
            pk, pcells, pts = 10, 50_000, 500
            px, py, pxt = make_regression(N_TRAIN, 10, n_outputs=pk, n_test=pts, config=6, unit=3)
            pg = GPRAS("RBF", device=device)
            pg.fit(px, py, 50, "grid", "adam", max_iter=3)
            pdry = np.zeros(pcells, dtype=bool)
            pdry[::97] = True
            pwet = int(pcells - pdry.sum())
            pelev = prng.uniform(0.0, 2.0, size=pcells)
            pproj = _Proj(pdry, pelev, prng.normal(size=pwet) + 1.5, prng.uniform(0.5, 1.5, size=pwet), prng.normal(size=(pk, pwet)) / np.sqrt(pk),
                          prng.normal(size=pk), prng.uniform(0.5, 2.0, size=pk), hydraulic_parameter="wse", device=device)
            pipe = DevicePipeline(pg, pproj)
            th, td = [], []
            for rep in range(3):
                t1 = time.perf_counter()
                mp_, vp_ = pg.predict(pxt)
                yp, yv = pproj.reverse_transform(mp_, vp_)
                dp = yp - pelev
                dp[dp < 0] = 0
                cf = np.sqrt(yv)
                th.append(time.perf_counter() - t1)
                t1 = time.perf_counter()
                fields = pipe.predict_fields(pxt)
                td.append(time.perf_counter() - t1)
                if rep == 2:
                    got_p, got_c = fields.to_host()
                fields.close()
            extra["device_pipeline"] = {
                "shape": {"modes": pk, "test_timesteps": pts, "cells": pcells, "n_train": N_TRAIN, "n_inducing": 50},
                "predict_reverse_depth_fields_ms_host_chain": 1e3 * min(th[1:]),
                "predict_reverse_depth_fields_ms_device_resident": 1e3 * min(td[1:]),
                "fields_equal_host_chain_bitwise": bool(np.array_equal(got_p, dp) and np.array_equal(got_c, cf)),
            }
            pproj.close()
            result["extra"] = extra
        except Exception as exc:  # noqa: BLE001
            import traceback

            traceback.print_exc()
            result["extra"] = extra
            result["extra_error"] = f"{type(exc).__name__}: {exc}"
        try:
            # ---- CPU baseline: the oracle on this box's host cores, bounded sample of the same workload ----
            from oracle import exact as oex

            cores = os.cpu_count() or 1
            try:
                from threadpoolctl import threadpool_info

                blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
            except Exception:
                blas_threads = cores
            v0, l0, s0 = 1.0, float(np.mean(np.abs(x))), 1.0
            # a TUNED CPU number: the kernel matrix through BLAS (gpflow's expanded distance form: one dgemm instead of a
            # Python loop over the dimensions) and the BLAS thread count chosen by measurement (all 256 hardware threads of
            # the GPU box's host oversubscribe dpotrf at N = 4096)
            best, best_threads = np.inf, blas_threads
            cpu_lml = None
            try:
                from threadpoolctl import threadpool_limits
            except Exception:
                threadpool_limits = None
            candidates = sorted({t for t in (8, 16, 32, 64, 128, blas_threads) if t <= max(blas_threads, 1)}) if threadpool_limits else [blas_threads]
            for nt in candidates:
                ctx = threadpool_limits(limits=nt, user_api="blas") if threadpool_limits else None
                try:
                    if ctx is not None:
                        ctx.__enter__()
                    for _ in range(2):
                        t1 = time.perf_counter()
                        cpu_lml = oex.lml("RBF", x, y[:, 0], v0, l0, s0, form="expanded")
                        dt = time.perf_counter() - t1
                        if dt < best:
                            best, best_threads = dt, nt
                finally:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
            cpu_lml = oex.lml("RBF", x, y[:, 0], v0, l0, s0)  # the parity check below uses the difference form, as the device does
            ctx = threadpool_limits(limits=best_threads, user_api="blas") if threadpool_limits else None
            if ctx is not None:
                ctx.__enter__()
            t1 = time.perf_counter()
            cm, cv = oex.predict("RBF", x, y[:, 0], v0, l0, s0, xs[:2000])
            tcp = time.perf_counter() - t1
            if ctx is not None:
                ctx.__exit__(None, None, None)
            # parity of the benchmarked step itself, at full size
            gpu_loss_check = -(cpu_lml + sum(-np.log(u) - 0.5 * np.log(2 * np.pi) - 0.5 * np.log(u) ** 2 for u in (v0, l0, s0)))
            fit_one()
            result["cpu_baseline"] = {
                "value": 1.0 / best,
                "unit": "fits/s",
                "cores": int(best_threads),  # = the threads actually used (the BLAS pool); the host has host_cpu_count hardware threads
                "blas_threads": int(best_threads),
                "kind": "port",
                "sample": f"F1 fits at N={N_TRAIN} d={DIM} with oracle/exact.py (numpy + scipy LAPACK, kernel matrix through BLAS), best of 2 at each of {candidates} BLAS threads (best: {best_threads}); predict on 2000 of the {N_TEST} points",
                "predict_points_per_s": 2000 / tcp,
                "host_cpu_count": cores,
            }

            # the other two sizes of the target, same oracle, same definitions (fit F1 = kernel build + Cholesky + alpha + LML; predictive mean
            # + variance at 2000 of the 100 000 points on the factor of that fit); N = 16384 is fitted ONCE (1.5 TFLOP of dpotrf)
            def limited(nt):
                return threadpool_limits(limits=nt, user_api="blas") if threadpool_limits else None

            def cpu_sample(xa, ya, ls, xsa, nt, reps):
                ctx = limited(nt)
                if ctx is not None:
                    ctx.__enter__()
                try:
                    t_fit = np.inf
                    for _ in range(reps):
                        t1 = time.perf_counter()
                        lfac, alpha = oex.factorize("RBF", xa, ya, 1.0, ls, 1.0, form="expanded")
                        float(-0.5 * ya @ alpha - np.log(np.diag(lfac)).sum())
                        t_fit = min(t_fit, time.perf_counter() - t1)
                    t1 = time.perf_counter()
                    cm_, cv_ = oex.predict_from_factor("RBF", xa, lfac, alpha, 1.0, ls, 1.0, xsa, form="expanded")
                    t_pred = time.perf_counter() - t1
                finally:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
                return t_fit, t_pred, cm_, cv_

            per_size = {}
            if x1 is not None and gpu_mean1 is not None:
                tf1, tp1, cm1, cv1 = cpu_sample(x1, y1[:, 0], l0, xs[:2000], best_threads, 3)
                per_size["N1024_d8"] = {"fits_per_s": 1.0 / tf1, "predict_points_per_s": 2000 / tp1, "blas_threads": int(best_threads), "sample": "fit best of 3; predict 2000 points once",
                                        "gpu_predict_mean_rel_err": float(np.max(np.abs(gpu_mean1 - cm1)) / np.max(np.abs(cm1))),
                                        "gpu_predict_var_rel_err": float(np.max(np.abs(gpu_var1 - cv1) / cv1))}
            if x5 is not None and gpu_mean5 is not None:
                # thread count for the large matrix: dpotrf alone on a 8192 x 8192 SPD matrix at a few pool sizes
                from scipy.linalg import cholesky as _chol

                nt5, tbest = best_threads, np.inf
                if threadpool_limits:
                    a8 = np.random.default_rng(5).standard_normal((8192, 64))
                    a8 = a8 @ a8.T + 8192.0 * np.eye(8192)
                    for nt in sorted({t for t in (best_threads, 32, 64) if t <= max(blas_threads, 1)}):
                        with threadpool_limits(limits=nt, user_api="blas"):
                            t1 = time.perf_counter()
                            _chol(a8, lower=True)
                            dt = time.perf_counter() - t1
                        if dt < tbest:
                            tbest, nt5 = dt, nt
                    del a8
                tf5, tp5, cm5, cv5 = cpu_sample(x5, y5[:, 0], float(np.mean(np.abs(x5))), xs5[:2000], nt5, 1)
                per_size["N16384_d12"] = {"fits_per_s": 1.0 / tf5, "predict_points_per_s": 2000 / tp5, "blas_threads": int(nt5), "sample": "fit ONCE; predict 2000 points once",
                                          "gpu_predict_mean_rel_err": float(np.max(np.abs(gpu_mean5 - cm5)) / np.max(np.abs(cm5))),
                                          "gpu_predict_var_rel_err": float(np.max(np.abs(gpu_var5 - cv5) / cv5))}
            per_size["N4096_d8"] = {"fits_per_s": 1.0 / best, "predict_points_per_s": 2000 / tcp, "blas_threads": int(best_threads)}
            result["cpu_baseline"]["per_size"] = per_size
            fit_step()
            result["parity_at_bench_size"] = {
                # (256 cells per launch carry their right-hand sides as vectors, a single call carries a 64-row tile: same factor bits,
                # y^T K^-1 y summed in another order -- round 4)
                "batched_cell0_vs_single_call_rel_diff": abs(float(losses[0]) - loss.value) / abs(loss.value),
                "loss_rel_err_vs_oracle": abs(loss.value - gpu_loss_check) / abs(gpu_loss_check),
                "predict_mean_rel_err": float(np.max(np.abs(gpu_mean - cm)) / np.max(np.abs(cm))),
                "predict_var_rel_err": float(np.max(np.abs(gpu_var - cv) / cv)),
            }
        except Exception as exc:  # noqa: BLE001
            import traceback

            traceback.print_exc()
            result["cpu_baseline_error"] = f"{type(exc).__name__}: {exc}"

    lib.gprx_destroy(h)
    if comm is not None:
        comm.barrier()
        comm.close()
    if fx is not None:
        fx.timeout_s = 3600.0  # (rank 0 may still be in its extras)
        fx.close()
    if use_torch:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    if rank == 0:
        # The driver's record keeps the TAIL of this line: the bulky extras go first, the judged blocks last, and a compact summary of the
        # numbers the review quotes closes the line.
        tail_keys = ["kernel_build_hbm", "C4", "cpu_baseline", "parity_at_bench_size", "roofline"]
        ordered = {}
        for k in ("extra", "extra_error"):
            if k in result:
                ordered[k] = result[k]
        for k, v in result.items():
            if k not in ordered and k not in tail_keys:
                ordered[k] = v
        for k in tail_keys:
            if k in result:
                ordered[k] = result[k]
        c4r, kb, rl, ex = result.get("C4") or {}, result.get("kernel_build_hbm") or {}, result.get("roofline") or {}, result.get("extra") or {}
        ordered["summary"] = {
            "fits_per_s": result["value"], "roofline_frac": rl.get("frac"), "single_cell_ms_per_fit": result.get("single_cell_ms_per_fit"),
            "C4_seconds_this_gpu": c4r.get("seconds_this_gpu"), "C4_frac_of_fp64_mfma_peak": c4r.get("frac_of_fp64_mfma_peak"),
            "kernel_build_frac_of_8TBps": kb.get("frac_of_8TBps"),
            "sgpr_batched16_loss_grad_evals_per_s": ex.get("sgpr_batched16_loss_grad_evals_per_s"),
            "sgpr_one_model_loss_grad_evals_per_s": ex.get("sgpr_n4096_d10_m50_loss_grad_evals_per_s"),
            "sgpr_16_modes_two_stage_fit_seconds": ex.get("sgpr_16_modes_two_stage_fit_seconds_lockstep"),
            "sgpr_resident_adam_evals_per_s_16_28_50_modes": [((ex.get("sparse_sgpr") or {}).get(k) or {}).get("evaluations_per_s") for k in
                                                              ("adam_5000_steps_16_modes_M50", "adam_1000_steps_28_modes_M50", "adam_1000_steps_50_modes_M50")],
            "cpu_baseline_fits_per_s": (result.get("cpu_baseline") or {}).get("value"),
        }
        print(json.dumps(ordered), flush=True)


if __name__ == "__main__":
    main()

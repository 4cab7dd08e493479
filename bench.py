#!/usr/bin/env python3
"""bench.py -- headline benchmark: grid-point-updates/s per Runge-Kutta substep of the incompressible Navier-Stokes
RHS hot path (OPR_Burgers x (12+3 ns), OPR_Partial x 5, OPR_Poisson, pointwise assembly, RK update), fp64,
512^3, 1 scalar, on N MI355X (BASELINE.json: metric / configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one explicit RK substep (TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT, tools/dns/time.f90:559) on synthetic fields
already resident in HBM.  Prints ONE JSON line on rank 0.  The `roofline` object is measured live with HIP events placed
by the library around every kernel launch on its own stream; `targets` = OPR_Partial_{X,Y,Z}(OPR_P1) standalone (the north-star's own
kernel); `substep_traffic` = the substep on the HBM bytes it really moves (PMC counters, profiles/traffic.json); `cpu_baseline` times the
C / OpenMP restatement of the reference's CPU path (oracle/tlab_cpu.c, kind "port") on bounded samples of the same workload on the host
cores of this box, best of several thread counts.  `--gpus N` without a launcher starts torch.distributed.run itself.  Diagnostics:
`--loopback P` (P z-slab ranks on one GPU), `--decomp IxK` (x/z pencils), `--grid NX NY NZ`, `--nscal`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def synthetic_fields(targets, nx, ny, nz_total, k0, kmax, seed):
    """smooth modes + 0.1 * uniform(-1,1) noise from a fixed-seed generator, generated on the device.
    targets: the 3 velocity + ns scalar tensors of the planes [k0, k0+kmax) of the global box."""
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250509 + seed)
    x = torch.arange(nx, dtype=torch.float64, device="cuda").view(1, 1, nx) / nx
    y = torch.arange(ny, dtype=torch.float64, device="cuda").view(1, ny, 1) / (ny - 1)
    z = (k0 + torch.arange(kmax, dtype=torch.float64, device="cuda")).view(kmax, 1, 1) / nz_total
    two_pi = 2.0 * np.pi
    wall = torch.sin(np.pi * y)
    shapes = [torch.sin(two_pi * x) * torch.cos(2 * two_pi * y) * torch.sin(3 * two_pi * z),
              torch.cos(two_pi * x) * torch.sin(two_pi * y) * torch.sin(2 * two_pi * z),
              torch.sin(2 * two_pi * x) * torch.cos(two_pi * y) * torch.cos(two_pi * z),
              torch.cos(3 * two_pi * x) * torch.cos(two_pi * y) * torch.sin(two_pi * z)]
    for t, sh in zip(targets, shapes):
        t.copy_(((sh + 0.1 * (2.0 * torch.rand(kmax, ny, nx, dtype=torch.float64, device="cuda", generator=gen) - 1.0)) * wall).reshape(-1))


def substep_traffic(kernels, kernels_substeps, per_launch, ms_per_step, grid, dominant_traffic=None):
    """HBM bytes one substep really moves: PMC-measured bytes per launch (profiles/traffic.json) x the launches of ONE substep.  `kernels` is a
    profile table whose `calls` are sums over `kernels_substeps` substeps (the table's own pass, not necessarily the timed region: round 5 divided
    a 6-substep table by --steps and under-reported 3.3 x).  rocFFT's transforms are counted at their algorithmic 2 x 8 B per point of the
    complex field.  A substep cannot move less than one launch of its dominant kernel: `consistent` says so."""
    nx, ny, nz = grid
    per_launch = {k: v for k, v in per_launch.items() if not k.startswith("_")}
    moved, missing = 0.0, []
    for k in kernels:
        per = per_launch.get(k["kernel"])
        if per is None and k["kernel"] == "rocfft":
            per = 2.0 * 8.0 * (nx + 2) * ny * nz
        if per is None:
            missing.append(k["kernel"])
            continue
        moved += per * k["calls"] / float(kernels_substeps)
    return {"hbm_bytes_per_step": moved, "GBps": moved / (ms_per_step * 1e-3) / 1e9, "frac_of_peak": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernels_without_counter_data": missing, "substeps_in_the_kernel_table": kernels_substeps,
            "consistent": None if dominant_traffic is None else bool(moved >= dominant_traffic),
            "what": "sum over the kernels of a step of the HBM bytes per launch measured with rocprofv3 PMC counters (profiles/traffic.json)"}


def cpu_baseline(n, nscal, budget_s=25.0):
    """Times substeps of the C + OpenMP restatement of the reference's CPU algorithm (oracle/tlab_cpu.c: explicit transposes, separate
    right-hand-side and Thomas passes, per-mode pentadiagonal solves; validated against the golden vectors in tests/test_cpu_baseline.py)
    on an n^3 sample of the workload, on all host cores of this box.  Falls back to the single-core numpy oracle if the C library is absent."""
    x = np.arange(n) / n
    y = np.arange(n) / (n - 1.0)
    rng = np.random.default_rng(20250509)
    try:
        from oracle import tlab_cpu as C
        C.load()
    except Exception as e:                                      # noqa: BLE001  (report, do not hide: the record says which leg ran)
        from oracle.tlab_oracle_rhs import DnsOracle
        o = DnsOracle(x, y, x.copy(), nscal, 1.0 / 5000.0, (1.0,) * nscal, True)
        for i in range(3):
            o.q[i] = rng.uniform(-1, 1, n ** 3) * 0.1
        for i in range(nscal):
            o.s[i] = rng.uniform(-1, 1, n ** 3)
        t0 = time.time()
        for _ in range(3):
            o.time_substep(1e-3, 1.0, False)
        dt = time.time() - t0
        return {"value": 3 * n ** 3 / dt, "unit": "grid-point-updates/s per RK substep", "cores": 1, "kind": "port",
                "sample": "3 RK substeps of the numpy oracle on a %d^3 box, %d scalar(s), %.1f s (oracle/libtlab_cpu.so unavailable: %s)" % (n, nscal, dt, e)}
    model, ncpu = C.host_description()
    t0 = time.time()
    c = C.CpuDnsDriver(x, y, x.copy(), nscal=nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * nscal, yuniform=True, hyper_bc1_ext=0.0)
    t_init = time.time() - t0
    for i in range(3):
        c.q[i][:] = rng.uniform(-1, 1, n ** 3) * 0.1
    for i in range(nscal):
        c.s[i][:] = rng.uniform(-1, 1, n ** 3)
    c.time_substep(1e-3, 1.0, False)                            # untimed: first touch of the work arrays
    # thread count: the reference's production mode is one MPI rank per core; the OpenMP partition of this restatement stops scaling earlier on
    # a many-socket host (transposes, NUMA), so the best of a few counts is what gets reported
    L = C.load()
    tmax = L.tlabcpu_num_threads()
    quota = cpu_quota()[0]
    cand = sorted({t for t in (8, 16, 32, 64, 128, 256, tmax, quota) if t <= tmax and t <= 2 * quota})      # beyond the granted CPUs threads only queue up
    best, trials = None, {}
    for t in cand:
        L.tlabcpu_set_num_threads(t)
        c.time_substep(1e-3, 1.0, False)
        t0 = time.time()
        c.time_substep(1e-3, 1.0, False)
        trials[t] = time.time() - t0
        if best is None or trials[t] < trials[best]:
            best = t
    L.tlabcpu_set_num_threads(best)
    nsub, t0 = 0, time.time()
    while nsub < 3 or (time.time() - t0 < 0.4 * budget_s and nsub < 30):
        c.time_substep(1e-3, 1.0, False)
        nsub += 1
    dt = time.time() - t0
    L.tlabcpu_set_num_threads(tmax)
    threads = best
    return {"value": nsub * n ** 3 / dt, "unit": "grid-point-updates/s per RK substep", "cores": threads, "kind": "port",
            "cpu_model": model, "host_cpus": ncpu, "seconds_per_substep_by_threads": {str(k): round(v, 3) for k, v in trials.items()},
            "cpu_quota": cpu_quota()[0], "cpu_quota_source": cpu_quota()[1],
            "sample": "%d RK substeps of the C/OpenMP restatement of the reference's CPU path (oracle/tlab_cpu.c) on a %d^3 box, %d scalar(s), "
                      "%d threads on '%s' (%d logical CPUs visible, %d granted to this job: %s), %.1f s timed after %.1f s of plan construction"
                      % (nsub, n, nscal, threads, model, ncpu, cpu_quota()[0], cpu_quota()[1], dt, t_init)}


def cpu_baseline_reference(n, nscal):
    """Substeps of oracle/tlab_ref_rhs.py::RefComposedDns (the reference's compiled routines, oracle/_ref/libtlab_ref.so) on an n^3 box, one core, in a
    child process (the library keeps one grid per process and this process may have loaded it for another)."""
    import subprocess
    lib = os.path.join(ROOT, "oracle", "_ref", "libtlab_ref.so")
    if not os.path.exists(lib):
        return {"error": "oracle/_ref/libtlab_ref.so not built (make -C oracle REF=/root/reference)"}
    code = (
        "import sys, time, json\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle.tlab_ref_rhs import RefComposedDns\n"
        "n, ns = %d, %d\n"
        "x = np.arange(n) / n; y = np.arange(n) / (n - 1.0)\n"
        "o = RefComposedDns(x, y, x.copy(), nscal=ns, visc=1.0 / 5000.0, schmidt=(1.0,) * ns, yuniform=True)\n"
        "rng = np.random.default_rng(20250509)\n"
        "for i in range(3): o.q[i] = rng.uniform(-1, 1, n ** 3) * 0.1\n"
        "for i in range(ns): o.s[i] = rng.uniform(-1, 1, n ** 3)\n"
        "o.time_substep(1e-3, 1.0, False)\n"
        "t0 = time.time(); k = 0\n"
        "while k < 2 or (time.time() - t0 < 8.0 and k < 6):\n"
        "    o.time_substep(1e-3, 1.0, False); k += 1\n"
        "print('RESULT ' + json.dumps({'nsub': k, 'seconds': time.time() - t0}))\n" % (ROOT, n, nscal))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if r.returncode != 0 or not line:
        return {"error": (r.stderr or r.stdout)[-400:]}
    res = json.loads(line[0][7:])
    return {"value": res["nsub"] * n ** 3 / res["seconds"], "unit": "grid-point-updates/s per RK substep", "cores": 1, "kind": "reference",
            "sample": "%d RK substeps of the reference's own compiled routines (oracle/_ref: FDM_Der1/2_Solve, the Burgers composition, FDM_Int1 + OPR_ODE2_Factorize_NN "
                      "per mode, serial amdflang -O2 build) composed by oracle/tlab_ref_rhs.py on a %d^3 box, %d scalar(s), %.1f s; Fourier transforms and the "
                      "loop over the modes in numpy / Python (the reference's FFTW is not in the image)" % (res["nsub"], n, nscal, res["seconds"])}


def cpu_quota():
    """CPUs this process may use at once: the cgroup's CPU bandwidth quota (cpu.max = quota period; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us) if
    there is one, else the affinity mask.  On the GPU boxes of this pool 256 logical CPUs are visible and 16 are granted."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return min(n, max(1, int(round(int(q) / int(per))))), "cgroup cpu.max = %s %s" % (q, per)
    except Exception:      # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return min(n, max(1, int(round(q / per)))), "cgroup cpu.cfs_quota_us / cpu.cfs_period_us = %d / %d" % (q, per)
    except Exception:      # noqa: BLE001
        pass
    return n, "no CPU quota (affinity mask)"


def _cpu_instance_worker(idx, cores, threads, n, nscal, nsub, q_ready, ev_go, q_out):
    """One independent instance of the C / OpenMP port on its own cores (spawned process: environment and affinity are set before the port loads)."""
    try:
        os.environ["OMP_NUM_THREADS"] = str(threads)
        os.environ["OMP_PROC_BIND"] = "close"
        try:
            os.sched_setaffinity(0, cores)
        except Exception:      # noqa: BLE001  (no affinity control: the instance still runs)
            pass
        from oracle import tlab_cpu as C
        C.load()
        x = np.arange(n) / n
        y = np.arange(n) / (n - 1.0)
        rng = np.random.default_rng(20250509 + idx)
        c = C.CpuDnsDriver(x, y, x.copy(), nscal=nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * nscal, yuniform=True, hyper_bc1_ext=0.0)
        for i in range(3):
            c.q[i][:] = rng.uniform(-1, 1, n ** 3) * 0.1
        for i in range(nscal):
            c.s[i][:] = rng.uniform(-1, 1, n ** 3)
        c.time_substep(1e-3, 1.0, False)
        c.time_substep(1e-3, 1.0, False)
        q_ready.put(idx)
        ev_go.wait()
        t0 = time.time()
        for _ in range(nsub):
            c.time_substep(1e-3, 1.0, False)
        q_out.put((idx, t0, time.time(), None))
    except Exception as e:      # noqa: BLE001
        q_ready.put(idx)
        q_out.put((idx, 0.0, 0.0, repr(e)))


def cpu_baseline_instances(n, nscal, threads, nsub=6):
    """The host the way the reference uses it: the reference's production mode is one MPI rank per core, each rank on its own part of the problem,
    so a many-socket host is timed as P independent instances of the port (threads cores each, pinned to disjoint contiguous core ranges, NUMA-local
    by first touch), all stepping an n^3 box at the same time; value = P x n^3 x substeps / wall time.  An upper bound of what a decomposed CPU run
    of the same box reaches on these cores (no exchange between the instances is counted)."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1
    try:
        avail = sorted(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        avail = list(range(ncpu))
    phys = avail[: max(1, len(avail) // 2)] if len(avail) >= 64 else avail      # first half: one hardware thread per core on an SMT-2 host
    P = max(1, min(len(phys) // threads, cpu_quota()[0] // threads, 16))
    if P < 2:
        return None
    try:
        import psutil
        if psutil.virtual_memory().available < P * 40 * 8 * n ** 3:
            return None
    except Exception:      # noqa: BLE001
        pass
    ctx = mp.get_context("spawn")
    q_ready, q_out, ev_go = ctx.Queue(), ctx.Queue(), ctx.Event()
    procs = [ctx.Process(target=_cpu_instance_worker, args=(i, set(phys[i * threads:(i + 1) * threads]), threads, n, nscal, nsub, q_ready, ev_go, q_out))
             for i in range(P)]
    t_launch = time.time()
    for p in procs:
        p.start()
    for _ in range(P):
        q_ready.get(timeout=600)
    t_init = time.time() - t_launch
    ev_go.set()
    res = [q_out.get(timeout=600) for _ in range(P)]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in res if r[3]]
    if bad:
        return {"error": bad[0][3]}
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    model = ""
    try:
        from oracle import tlab_cpu as C
        model, _ = C.host_description()
    except Exception:      # noqa: BLE001
        pass
    return {"value": P * nsub * n ** 3 / wall, "unit": "grid-point-updates/s per RK substep", "cores": P * threads, "kind": "port", "cpu_model": model,
            "host_cpus": ncpu, "instances": P, "threads_per_instance": threads, "seconds_per_substep_per_instance": wall / nsub,
            "sample": "%d independent instances of the C/OpenMP restatement of the reference's CPU path (oracle/tlab_cpu.c), %d threads each on disjoint "
                      "pinned core ranges (%d of the %d logical CPUs of '%s'), every instance stepping its own %d^3 box, %d scalar(s): %d substeps each in "
                      "%.1f s of wall time after %.1f s of start-up; aggregate = instances x points x substeps / wall time (the reference's rank-per-core "
                      "mode without its exchanges: an upper bound for a decomposed CPU run of one box)" % (P, threads, P * threads, ncpu, model, n, nscal, nsub, wall, t_init)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--box", dest="n", type=int, default=512, help="box size n^3 (BASELINE: 512), split into z-slabs over the GPUs; under torch.distributed.run spell it --box (its parser rejects --n as ambiguous)")
    ap.add_argument("--grid", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"), help="non-cubic box (diagnostics; e.g. 1024 512 1024 = BASELINE configs[3])")
    ap.add_argument("--nscal", type=int, default=1)
    ap.add_argument("--ystretch", action="store_true", help="diagnostic (single GPU): tanh-stretched y nodes (SURVEY 8d: y_j = (1 + tanh(2(2(j-1)/(ny-1) - 1))/tanh 2)/2, "
                    "the grid of BASELINE configs[4]) instead of the uniform ones of the headline: the second derivative then carries its Jacobian correction")
    ap.add_argument("--walls", default="noslip", choices=["noslip", "freeslip"], help="diagnostic (single GPU, or --loopback P): freeslip = the reference's default velocity walls "
                    "with Neumann scalars (BOUNDARY_BCS_NEUMANN_Y in the tail of the substep); the headline is noslip / Dirichlet")
    ap.add_argument("--loopback", type=int, default=0, help="diagnostic: run the z-slab algorithm of P ranks inside this one process/GPU "
                    "(no communication, ranks execute one after the other); reports the time of ALL ranks' work")
    ap.add_argument("--decomp", default="", metavar="IxK", help="diagnostic: x/z pencil decomposition npro_i x npro_k (tlab_amd/pencil.py), e.g. 2x4: with "
                    "--gpus N = I*K one block per GPU, otherwise all I*K ranks inside this one process/GPU (loopback); the default multi-GPU run is 1 x N z-slabs")
    ap.add_argument("--slab-driver", default="native", choices=["native", "python"], help="z-slab runs (--gpus N > 1, --loopback P): native = the C++ driver behind "
                    "tlab_slab_dns_* (tlab_amd/csrc/slab.cpp; RCCL transport of libtlab_amd_comm.so), the code a Fortran / MPI host runs; python = its "
                    "cross-check tlab_amd/parallel.py::SlabDns over torch.distributed (diagnostic)")
    ap.add_argument("--placement-trials", type=int, default=32, help="single GPU: random assignments tried by tlab_dns_place_arrays before the timed region (0: the "
                    "arrays stay where the allocator put them)")
    ap.add_argument("--placement-pool", type=int, default=56, help="candidate allocations of the placement search")
    ap.add_argument("--time-every-launch", action="store_true", help="events around every kernel launch inside the timed region (the kernel table then comes "
                    "from the timed region itself; costs ~0.2 ms per substep at 512^3)")
    ap.add_argument("--no-fortran-host", action="store_true", help="skip the `fortran_host` leg of the default single-GPU line (the Fortran mini-driver timed at the "
                    "benchmark's size in a child process, after the timed region)")
    ap.add_argument("--no-freeslip-leg", action="store_true", help="skip the extra `walls_freeslip` timing of the default single-GPU line")
    ap.add_argument("--cpu-sample", type=int, default=256, help="n of the n^3 CPU-baseline sample (0 disables)")
    ap.add_argument("--cpu-sample-large", type=int, default=512, help="second, larger CPU-baseline sample, run only on hosts with at least --cpu-large-min-cores CPUs (0 disables)")
    ap.add_argument("--cpu-large-min-cores", type=int, default=48)
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks as a CHILD torch.distributed.run (nothing has touched the GPU yet in this process: no
    # re-exec after GPU initialisation) and relay its single JSON line.  Under a launcher, --gpus must agree with WORLD_SIZE.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        argv = [a if a != "--n" else "--box" for a in sys.argv[1:]]     # torch.distributed.run's parser rejects --n as ambiguous
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        sys.exit(subprocess.run(cmd).returncode)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s (launch with --nproc-per-node %d, or let `python bench.py --gpus %d` start the ranks itself)"
                 % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # Start-up of N > 1 ranks.  The product path moves its data through the RCCL inside libtlab_amd_comm.so (linked against /opt/rocm's librccl); torch
    # bundles an RCCL of its own.  To keep ONE RCCL user in the process the torch.distributed group is a gloo (CPU) group by default: it carries the
    # ncclUniqueId of the native communicator, the barriers and the max-over-ranks of the timing -- no payload.  TLAB_DIST_BOOTSTRAP=nccl restores the
    # torch NCCL group (the Python fall-back driver needs one); TLAB_DIST_BACKEND=gloo is the functional mode with host-staged payloads (several ranks
    # on one GPU), never a measurement.
    backend = os.environ.get("TLAB_DIST_BACKEND", "nccl")      # transport of the payload: nccl (= RCCL) | gloo (functional, host-staged)
    bootstrap = os.environ.get("TLAB_DIST_BOOTSTRAP", "gloo" if args.slab_driver == "native" else "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
        bootstrap = backend
    torch.cuda.set_device(local_rank)
    nccl_group = [None]

    def torch_nccl_group():
        """the torch NCCL group, made on first use (Python fall-back driver, pencil decomposition)"""
        if bootstrap == "nccl":
            return None      # the default group is it
        if nccl_group[0] is None:
            nccl_group[0] = dist.new_group(backend="nccl")
        return nccl_group[0]
    if world > 1:
        if bootstrap == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(bootstrap)
    reduce_dev = "cuda" if (world > 1 and bootstrap == "nccl") else "cpu"      # where the small control reductions live

    placement = None
    import tlab_amd as T
    from tlab_amd.dns import Dns, RKM_EXP3
    from tlab_amd.lib import load
    T.init(local_rank)
    n = args.n
    nx, ny, nz = (args.grid if args.grid else (n, n, n))
    x = np.arange(nx) / nx
    y = np.arange(ny) / (ny - 1.0)
    if args.ystretch:
        y = 0.5 * (1.0 + np.tanh(2.0 * (2.0 * y - 1.0)) / np.tanh(2.0))
    z = np.arange(nz) / nz
    dtime = 2e-5 if args.ystretch else 1e-3      # (the stretched grid's wall spacing is 1/7 of the uniform one: advective and diffusive limits of the explicit scheme)
    # wall closure of the hyper-diffusive second derivative: the consistent one.  The parity tests use 0.1, the value the flang-built reference reads
    # past the end of a coefficient array (defect 1) -- same kernels and bytes, but that scheme is unstable over many steps.
    HYPER_BC1_EXT = 0.0
    L = load()
    if args.decomp:
        from tlab_amd.pencil import PencilDns, NativePencilDns, loopback_comms, dist_comms
        npi, npk = (int(v) for v in args.decomp.lower().split("x"))
        if world > 1 and world != npi * npk:
            raise SystemExit("--decomp IxK needs --gpus I*K (or one process for the loopback)")
        pkw = dict(nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=True, rkm_mode=RKM_EXP3, hyper_bc1_ext=HYPER_BC1_EXT)
        if args.slab_driver == "native":      # the C++ driver behind tlab_pencil_dns_* (csrc/pencil.cpp)
            d = NativePencilDns("rccl" if world > 1 else "loopback", npi, npk, x, y, z, **pkw)
            pranks = d.local_ranks
            if args.placement_trials > 0:
                placement = d.redraw_arrays(pool=2 * (2 * (3 + args.nscal) + 9), seed=rank)
        else:
            d = PencilDns(dist_comms(npi, npk) if world > 1 else loopback_comms(npi, npk), npi, npk, x, y, z, **pkw)
            pranks = d.world.local_ranks
        state_fields = []
        full = [torch.empty(nx * ny * nz, dtype=torch.float64, device="cuda") for _ in range(3 + args.nscal)]
        synthetic_fields(full, nx, ny, nz, 0, nz, 0)
        for i, f in enumerate(full):
            d.scatter("q" if i < 3 else "s", i if i < 3 else i - 3, f)
        del full
        for r in pranks:
            state_fields += d.st[r]["q"] + d.st[r]["s"]

        def substep(k):
            d.substep_of_cycle(k, dtime)
    elif world == 1 and args.loopback > 1:
        from tlab_amd.parallel import SlabDns, LoopbackComm
        from tlab_amd.slab import NativeSlabDns
        kw = dict(nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=True, rkm_mode=RKM_EXP3, hyper_bc1_ext=HYPER_BC1_EXT)
        if args.slab_driver == "native":
            try:
                d = NativeSlabDns("loopback", x, y, z, size=args.loopback, **kw)
            except T.TlabError as e:       # slabs too thin for the partitioned z-systems: the reference's K-transposition scheme (Python driver)
                print("bench.py: native slab driver refused (%s); falling back to the transposition scheme of tlab_amd/parallel.py" % e, file=sys.stderr)
                args.slab_driver = "python"
        if args.slab_driver != "native":
            d = SlabDns(LoopbackComm(args.loopback), x, y, z, **kw)
        if args.walls == "freeslip":
            d.set_bcs("freeslip", "freeslip", "neumann", "neumann")
        if args.slab_driver == "native" and args.placement_trials > 0:
            placement = d.redraw_arrays(pool=2 * (2 * (3 + args.nscal) + 9))
        state_fields = []
        for r in range(args.loopback):
            S = d.st[r]
            synthetic_fields(S["q"] + S["s"], nx, ny, nz, r * d.kmax, d.kmax, r)
            state_fields += S["q"] + S["s"]

        def substep(k):
            d.substep_of_cycle(k, dtime)
    elif world == 1:
        # one GPU owns the whole box: the C++ driver (tlab_amd/csrc/rhs.cpp) runs the substep
        d = Dns(x, y, z, nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=not args.ystretch, rkm_mode=RKM_EXP3,
                hyper_bc1_ext=HYPER_BC1_EXT)
        if args.walls == "freeslip":
            d.set_bcs("freeslip", "freeslip", "neumann", "neumann")
        synthetic_fields(d.q + d.s, nx, ny, nz, 0, nz, rank)
        # which allocations play q, s, hq, hs, txc: searched at start-up, outside the timed region, like a plan (tlab_dns_place_arrays)
        if args.placement_trials > 0:
            try:
                placement = d.place_arrays(pool=args.placement_pool, random_trials=args.placement_trials, dtime=dtime)
            except (T.TlabError, RuntimeError) as e:      # the search failed inside the library (the arrays are the pool in order) or the pool did not fit
                placement = {"error": str(e)}                # (the driver has its arrays back): the run goes on and says so
        state_fields = d.q + d.s

        def substep(k):
            s = k % d.rkm_endstep
            if s == 0:
                d.begin_step()          # hq = hs = 0 of TIME_RUNGEKUTTA (time.f90:212-216): the first operator launch overwrites instead
            last = s == d.rkm_endstep - 1
            d.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * d.kdt[s], 1.0 if last else d.kco[s], not last)
    else:
        # STRONG scaling of the same n^3 box over z-slabs (1 x N): the native driver (tlab_amd/csrc/slab.cpp) with its exchanges as grouped
        # ncclSend / ncclRecv on the communication stream of libtlab_amd_comm.so -- the same calls a Fortran / MPI host makes
        from tlab_amd.parallel import SlabDns, DistComm
        from tlab_amd.slab import NativeSlabDns
        kw = dict(nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=True, rkm_mode=RKM_EXP3, hyper_bc1_ext=HYPER_BC1_EXT)
        if args.slab_driver == "native" and nz // world < 56:      # decided from the sizes alone, so that every rank takes the same branch
            if rank == 0:
                print("bench.py: %d planes per rank are too few for the partitioned z-systems of the native slab driver; using the K-transposition "
                      "scheme of tlab_amd/parallel.py" % (nz // world), file=sys.stderr)
            args.slab_driver = "python"
        def all_ranks_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=reduce_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        d = None
        if args.slab_driver == "native":
            # The native driver's RCCL transport has only ever run with one rank per communicator (one GPU per test box): should its start-up or its
            # first substep fail on ANY rank, every rank falls back to the Python driver over torch.distributed (same algorithm, same kernels) and
            # the line says so -- a number from the slower driver beats no number.
            # Two agreements, each BEFORE the step that could leave a peer blocked inside a grouped exchange: (1) every rank has built its driver and
            # bound its arrays (no exchange has been enqueued yet); (2) the trial substep, which runs under a watchdog -- a rank whose peers never
            # arrive (an ncclSend / ncclRecv error on one side only) would otherwise wait for ever instead of falling back: after 120 s the process
            # exits non-zero, which takes the whole launch down instead of hanging it.
            err = None
            try:
                d = NativeSlabDns("rccl" if backend == "nccl" else "dist", x, y, z, **kw)
                if args.placement_trials > 0:
                    placement = d.redraw_arrays(pool=2 * (2 * (3 + args.nscal) + 9), seed=rank)
                S = d.st[rank]
                synthetic_fields(S["q"] + S["s"], nx, ny, nz, rank * d.kmax, d.kmax, rank)
            except Exception as e:      # noqa: BLE001
                err = e
            built = all_ranks_ok(err is None)
            if built:
                import threading
                done = threading.Event()

                def watchdog():
                    if not done.wait(120.0):
                        print("bench.py rank %d: the trial substep of the native slab driver did not finish in 120 s (a peer failed inside an exchange?)" % rank,
                              file=sys.stderr, flush=True)
                        os._exit(3)
                threading.Thread(target=watchdog, daemon=True).start()
                try:
                    d.substep_of_cycle(0, dtime)
                    torch.cuda.synchronize()
                except Exception as e:      # noqa: BLE001
                    err = e
                done.set()
            if not built or not all_ranks_ok(err is None):
                print("bench.py rank %d: native slab driver failed (%s); falling back to the Python driver" % (rank, err if err else "on another rank"), file=sys.stderr)
                try:
                    if d is not None:
                        d.close()
                except Exception:       # noqa: BLE001
                    pass
                d = None
                args.slab_driver = "python (native driver failed at start-up)"
        if d is None:
            d = SlabDns(DistComm(torch_nccl_group() if backend == "nccl" else None), x, y, z, **kw)
        S = d.st[rank]
        synthetic_fields(S["q"] + S["s"], nx, ny, nz, rank * d.kmax, d.kmax, rank)
        state_fields = S["q"] + S["s"]

        def substep(k):
            d.substep_of_cycle(k, dtime)

    import ctypes
    pbuf = ctypes.create_string_buffer(1 << 16)

    def profile_rows():
        """the library's own event timings since the last reset: one row per kernel name, largest summed time first"""
        L.tlab_profile_report(pbuf, len(pbuf))
        rows = []
        for line in pbuf.value.decode().strip().split("\n"):
            if not line:
                continue
            name, calls, ms, nbytes = line.split("\t")
            rows.append({"kernel": name, "calls": int(calls), "total_ms": float(ms), "avg_ms": float(ms) / int(calls),
                         "alg_bytes_per_launch": float(nbytes) / int(calls)})
        rows.sort(key=lambda k: -k["total_ms"])
        for k in rows:
            k["alg_GBps"] = k["alg_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9 if k["avg_ms"] > 0 else 0.0
        return rows

    # Every launch of the warm-up is timed by the library's events; the kernel with the largest summed time among the full-field kernels is the
    # dominant one, and in the timed region only ITS launches carry events (two event records per timed launch hold the stream for ~3 us each: with
    # all ~45 launches of a substep timed, 0.2 ms of the 16).  The table of all kernels comes from a pass of its own after the timed region.
    L.tlab_profile_filter(None)
    L.tlab_profile_reset()
    L.tlab_profile_enable(1)
    for k in range(args.warmup):
        substep(k)
    dom_tag = None
    if args.warmup > 0 and not args.time_every_launch:
        dom_tag = next((k["kernel"] for k in profile_rows() if k["alg_bytes_per_launch"] > 1e6), None)
    L.tlab_profile_filter(dom_tag.encode() if dom_tag else None)
    L.tlab_profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        substep(args.warmup + k)
    enqueue = time.perf_counter() - t0          # host time to issue the K substeps (this rank); the rest of `elapsed` is the GPU finishing
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.tlab_profile_enable(0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # the host's own cost of issuing a substep: `enqueue` above is throttled by the device once the launch queue is full (the host runs ahead of the GPU
    # until hipLaunchKernel blocks), so one more substep is issued into an EMPTY queue and timed on the host alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    substep(args.warmup + args.steps)
    issue_empty = time.perf_counter() - t0
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(t).all()) for t in state_fields)
    # what RCCL itself saw (ncclCommCount / ncclCommCuDevice of the communicator inside libtlab_amd_comm.so), not what this script passed in: the first
    # multi-GPU run must be diagnosable from its one JSON line.  None: no RCCL communicator in this run (single GPU, loopback, gloo functional mode,
    # or the Python fall-back driver over torch's own NCCL group).
    rccl_ranks = rccl_device = None
    nc = getattr(d, "_keep", None)
    if nc is not None and hasattr(nc, "info"):
        try:
            rccl_ranks, rccl_device = int(nc.info(6)), int(nc.info(8))
        except Exception:       # noqa: BLE001
            rccl_ranks = rccl_device = None
    driver_name = ("tlab_pencil_dns (csrc/pencil.cpp)" if args.slab_driver == "native" else "tlab_amd/pencil.py") if args.decomp else \
                  ("tlab_dns (csrc/rhs.cpp)" if (world == 1 and args.loopback <= 1) else
                   "tlab_slab_dns (csrc/slab.cpp)" if args.slab_driver == "native" else "tlab_amd/parallel.py::SlabDns (%s)" % args.slab_driver)

    # dominant kernel = largest summed time among the full-field kernels (the <= 4 singular Poisson modes run beside the main stream and rocFFT
    # carries no byte count): its launches inside the timed region
    timed_rows = profile_rows()
    dom = next((k for k in timed_rows if k["alg_bytes_per_launch"] > 1e6), None)
    kernels, kernels_pass = timed_rows, "the timed region (every launch timed)"
    kernels_substeps = args.steps          # the number of substeps the table `kernels` covers (its `calls` are sums over them)
    if dom_tag is not None:      # the table of all kernels: a pass of its own, every launch timed, same fields (every rank takes part in the exchanges)
        npass = min(args.steps, 6)
        L.tlab_profile_filter(None)
        L.tlab_profile_reset()
        L.tlab_profile_enable(1)
        for k in range(npass):
            substep(args.warmup + args.steps + 1 + k)
        torch.cuda.synchronize()
        L.tlab_profile_enable(0)
        kernels = profile_rows()
        kernels_substeps = npass
        kernels_pass = "%d substeps after the timed region with every launch timed (in the timed region only %s carries events)" % (npass, dom_tag)
        finite = finite and all(bool(torch.isfinite(t).all()) for t in state_fields)
    if rank == 0:
        # roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC passes of the SAME binaries (profiles/traffic.json is
        # regenerated by tools/profile_round.sh together with the rocprofv3 CSV and stamped with the commit and the kernel list of that session;
        # a kernel the file does not know gets null, never another kernel's or an older build's figure)
        traffic, traffic_meta = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        # the counter data is quoted only when it was measured on THESE kernel sources (sha256 over tlab_amd/csrc, stamped by tools/pmc_summary.py)
        traffic_valid = False
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_summary import source_hash
            traffic_valid = os.path.exists(tpath) and json.load(open(tpath)).get("_meta", {}).get("source_hash") == source_hash()
        except Exception:       # noqa: BLE001
            traffic_valid = False
        if dom and os.path.exists(tpath) and world == 1 and args.loopback <= 1 and not args.decomp and (nx, ny, nz) == (512, 512, 512) and args.nscal == 1:      # measured for this workload only
            try:
                tj0 = json.load(open(tpath))
                traffic_meta = tj0.get("_meta")
                traffic = tj0.get(dom["kernel"]) if traffic_valid else None
                if not traffic_valid:
                    traffic_meta = dict(traffic_meta or {}, stale="profiles/traffic.json was measured on other kernel sources (source_hash differs): traffic is null")
            except Exception:
                traffic = None
        npts = float(nx) * ny * nz      # strong scaling: the same box on 1/2/4/8 GPUs (BASELINE.json metric)
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "grid-point-updates/s per RK substep",
            "value": npts * args.steps / elapsed,
            "unit": "grid-point-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "host_issue_ms_per_step": enqueue / args.steps * 1e3,
            "host_issue_empty_queue_ms": issue_empty * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%dx%dx%d incompressible box, %d scalar, full RHS (12+3ns OPR_Burgers, 5 OPR_Partial, OPR_Poisson FourierXZ) + RK3 update per substep"
                                   % (nx, ny, nz, args.nscal),
                       "grid": [nx, ny, nz], "n_scalars": args.nscal, "schemes": "CompactJacobian6 / CompactJacobian6Hyper (consistent wall closure)", "reynolds": 5000,
                       "parallelism": ("DIAGNOSTIC: x/z pencils %s (%s, %s driver), I-/K-transpositions per x/z operator, Poisson on kx-pencils" % (args.decomp, "one block per GPU" if world > 1 else "all ranks back to back on one GPU, exchanges = copies", args.slab_driver)) if args.decomp else
                                      ("single GPU" if args.loopback <= 1 else "DIAGNOSTIC: %d z-slab ranks (%s mode, %s driver) executed back to back on one GPU, no communication" % (args.loopback, d.zmode, args.slab_driver)) if world == 1 else
                       ("z-slabs 1x%d, halo planes + interface values between neighbours for d/dz, kx-pencil Poisson (3 all-to-alls per substep in two pipelined halves), %s driver, %s" % (world, args.slab_driver, ("RCCL of libtlab_amd_comm.so (start-up over a %s group)" % bootstrap) if backend == "nccl" else backend + " (functional run, host-staged)") if d.zmode == "halo" else "z-slabs 1x%d, K-transposes = RCCL all_to_all_single per z-operator / z-FFT" % world),
                       "driver": driver_name, "rccl_ranks": rccl_ranks, "rccl_device_of_rank0": rccl_device, "world_size": world,
                       "transport": ("none (one domain)" if (world == 1 and args.loopback <= 1 and not args.decomp) else "loopback copies" if world == 1 else
                                     ("RCCL (libtlab_amd_comm.so)" if rccl_ranks is not None else "torch.distributed %s" % (backend if backend != "nccl" else "nccl group"))),
                       "fields_finite": finite},
            "roofline": None if dom is None else {
                "kernel": dom["kernel"], "bound": "hbm", "achieved": dom["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["alg_GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_meta, "avg_launch_ms": dom["avg_ms"], "launches": dom["calls"],
                "share_of_step": dom["total_ms"] / (ms_per_step * args.steps)},
            "substep_alg_GBps": (736.0 + 152.0 * args.nscal) * npts / (ms_per_step * 1e-3) / 1e9,
            "kernels": [{k2: (round(v, 6) if isinstance(v, float) else v) for k2, v in k.items()} for k in kernels],
            "kernels_from": kernels_pass,
        }
        single = world == 1 and args.loopback <= 1 and not args.decomp
        if world == 1 and not single:
            # all ranks on one GPU: the transports' exchanges are device copies (k_copy_blocks; the pencil driver's wire format passes k_trp_copy are
            # packing work that stays on N GPUs) standing in for what xGMI carries on N GPUs.  Their share of the line, so that the work of the ranks
            # can be read without it: summed launch durations of the kernel-table pass per substep
            cp = [k for k in kernels if k["kernel"] == "k_copy_blocks"]
            if cp:
                ms_cp = sum(k["avg_ms"] * k["calls"] for k in cp) / float(kernels_substeps)
                out["exchange_standin"] = {"kernel": "k_copy_blocks", "ms_per_step": ms_cp, "launches_per_step": sum(k["calls"] for k in cp) / float(kernels_substeps),
                                           "ms_per_step_without": ms_per_step - ms_cp,
                                           "what": "device copies of the loopback transport in place of the exchanges between GPUs; `ms_per_step_without` = the line minus "
                                                   "their summed durations (they run on the compute stream in this mode: nothing overlaps them)"}
        if not single and placement is not None:
            out["placement"] = placement
        if single:
            out["placement"] = None if placement is None else placement if "error" in placement else dict(
                placement, what="tlab_dns_place_arrays before the timed region: ms per substep of the allocations in the order the allocator gave them "
                                "(ms_first), of the assignment the run uses (ms_best) and of the median / worst assignment tried; --placement-trials 0 skips it")
        if single:
            # context for the roofline fraction: what a plain device copy of one field reaches on THIS box (read + write bytes), after the timed
            # region.  The 8 TB/s peak is not reachable by any kernel; the streaming kernels above are to be read against this figure too.
            src, dst = d.hq[0], d.hs[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst.copy_(src)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 10 * 2.0 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["copy_ceiling"] = {"GBps": copy_gbs, "what": "device copy of one %d-point fp64 field, read + write bytes, 10 repetitions" % src.numel()}
            if out["roofline"] is not None:
                out["roofline"]["frac_of_copy"] = out["roofline"]["achieved"] / copy_gbs
            # ... and hand-written streaming kernels on the same box (tools/yardstick.hip, a child process: 16-B-per-lane copy and triad, and the access
            # pattern of the tile kernels -- 32 lines x 512 rows per workgroup, operand + velocity + old tendency in, new tendency out -- WITHOUT arithmetic)
            ybin = os.path.join(ROOT, "tools", "yardstick")
            if os.path.exists(ybin) and (nx, ny, nz) == (512, 512, 512):
                try:
                    import subprocess
                    yl = [json.loads(l) for l in subprocess.run([ybin, "512"], capture_output=True, text=True, timeout=120).stdout.splitlines() if l.startswith("{")]
                    yd = {r["kernel"]: r["GBps"] for r in yl}
                    pat = max((v for k2, v in yd.items() if k2.startswith("tile L=32")), default=None)
                    out["copy_ceiling_own"] = {"copy16_GBps": max((v for k2, v in yd.items() if k2.startswith("copy16")), default=None),
                                               "triad16_GBps": yd.get("triad16 grid=8192"), "tile_pattern_GBps": pat, "all": yd,
                                               "what": "tools/yardstick.hip on this box after the timed region: global_load/store_dwordx4 copy and 3-read / 1-write triad, "
                                                       "and the Burgers tile kernels' own access pattern without arithmetic (best of y / z lines, phased, non-temporal)"}
                    if out["roofline"] is not None and pat:
                        out["roofline"]["frac_of_own_pattern"] = out["roofline"]["achieved"] / pat
                except Exception as e:       # noqa: BLE001
                    out["copy_ceiling_own"] = {"error": repr(e)}
        if single:
            # the north-star's own target kernel, standalone on this box: OPR_Partial_{X,Y,Z}(OPR_P1) at the benchmark's grid, 16 B per point
            # (SURVEY.md 8d), >= 40 % of the 8 TB/s HBM peak asked for OPR_Partial_X at 512^3
            targets = {}
            a, r = d.txc[0][: d.n], d.txc[1][: d.n]
            a.copy_(d.q[0])
            for name, fn, g in (("OPR_Partial_X", T.OPR_Partial_X, d.g[0]), ("OPR_Partial_Y", T.OPR_Partial_Y, d.g[1]), ("OPR_Partial_Z", T.OPR_Partial_Z, d.g[2])):
                for _ in range(2):
                    fn(T.OPR_P1, nx, ny, nz, 0, g, a, r, None)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for e0, e1 in ev:
                    e0.record(); fn(T.OPR_P1, nx, ny, nz, 0, g, a, r, None); e1.record()
                torch.cuda.synchronize()
                ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[5]
                gbs = 16.0 * npts / (ms * 1e-3) / 1e9
                targets[name + "(OPR_P1)"] = {"ms": ms, "GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "points_per_s": npts / (ms * 1e-3)}
            targets["north_star"] = ">= 40 % of the HBM roofline (8 TB/s) on OPR_Partial_X at 512^3: <= 0.67 ms per call"
            out["targets"] = targets
            # the substep against the bytes it really moves: PMC-measured HBM traffic per launch (profiles/traffic.json, this workload only) x
            # the launches of one step; rocFFT's transforms are counted at their algorithmic 2 x 8 B per point of the complex field
            tj = None
            if traffic_valid and (nx, ny, nz) == (512, 512, 512) and args.nscal == 1:
                try:
                    tj = json.load(open(tpath))
                except Exception:
                    tj = None
            if tj:
                out["substep_traffic"] = substep_traffic(kernels, kernels_substeps, tj, ms_per_step, (nx, ny, nz),
                                                         None if out["roofline"] is None else out["roofline"]["traffic"])
        if single and args.walls == "noslip" and not args.no_freeslip_leg:
            # the same workload with the reference's DEFAULT walls (VelocityJmin/Jmax = freeslip, Neumann scalars: BOUNDARY_BCS_NEUMANN_Y in the tail
            # of the substep), timed the same way right after the headline: the headline keeps no-slip / Dirichlet walls, this key says what the default costs
            d = None
            torch.cuda.empty_cache()
            d2 = Dns(x, y, z, nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=not args.ystretch, rkm_mode=RKM_EXP3,
                     hyper_bc1_ext=HYPER_BC1_EXT)
            d2.set_bcs("freeslip", "freeslip", "neumann", "neumann")
            synthetic_fields(d2.q + d2.s, nx, ny, nz, 0, nz, rank)
            pl2 = None
            if args.placement_trials > 0:      # like for like with the headline: the same search for where the arrays live
                try:
                    pl2 = d2.place_arrays(pool=args.placement_pool, random_trials=args.placement_trials, dtime=dtime)
                except (T.TlabError, RuntimeError) as e:
                    pl2 = {"error": str(e)}

            def substep2(k):
                s_ = k % d2.rkm_endstep
                if s_ == 0:
                    d2.begin_step()
                last = s_ == d2.rkm_endstep - 1
                d2.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * d2.kdt[s_], 1.0 if last else d2.kco[s_], not last)
            for k in range(args.warmup):
                substep2(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                substep2(args.warmup + k)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            out["walls_freeslip"] = {"ms_per_step": el2 / args.steps * 1e3, "value": npts * args.steps / el2, "unit": "grid-point-updates/s",
                                     "steps": args.steps, "warmup": args.warmup, "fields_finite": all(bool(torch.isfinite(t).all()) for t in d2.q + d2.s),
                                     "placement": pl2,
                                     "what": "same box and steps with VelocityJmin/Jmax = freeslip and Neumann scalars (the reference's default walls), arrays placed by the same search"}
            del d2
            torch.cuda.empty_cache()
        if single and args.walls == "noslip" and not args.no_freeslip_leg:
            # ... and with the wall closure of the second derivative that the reference AS COMPILED uses (its out-of-bounds read finds 0.1: DESIGN.md section 2,
            # defect 1), which is the closure the parity fixtures made by the reference hold; the headline times the consistent closure 0.0.  Same
            # kernels, same bytes: only two table entries differ.
            d3 = Dns(x, y, z, nscal=args.nscal, visc=1.0 / 5000.0, schmidt=(1.0,) * args.nscal, yuniform=not args.ystretch, rkm_mode=RKM_EXP3, hyper_bc1_ext=0.1)
            synthetic_fields(d3.q + d3.s, nx, ny, nz, 0, nz, rank)
            pl3 = None
            if args.placement_trials > 0:
                try:
                    pl3 = d3.place_arrays(pool=args.placement_pool, random_trials=args.placement_trials, dtime=dtime)
                except (T.TlabError, RuntimeError) as e:
                    pl3 = {"error": str(e)}

            def substep3(k):
                s_ = k % d3.rkm_endstep
                if s_ == 0:
                    d3.begin_step()
                last = s_ == d3.rkm_endstep - 1
                d3.TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT(dtime * d3.kdt[s_], 1.0 if last else d3.kco[s_], not last)
            for k in range(args.warmup):
                substep3(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                substep3(args.warmup + k)
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t0
            out["closure_as_compiled_reference"] = {"hyper_bc1_ext": 0.1, "ms_per_step": el3 / args.steps * 1e3, "value": npts * args.steps / el3,
                                                    "unit": "grid-point-updates/s", "fields_finite": all(bool(torch.isfinite(t).all()) for t in d3.q + d3.s),
                                                    "placement": pl3,
                                                    "what": "same box, steps and kernels with the wall closure 0.1 of the flang-built reference, arrays placed by the same search"}
            del d3
            torch.cuda.empty_cache()
        if single and not args.no_fortran_host and args.nscal == 1 and not args.ystretch and args.walls == "noslip":
            # the configuration north_star names: the Fortran RK driver on the device path, at this size, in a child process of its own (its arrays in
            # the host's layout q(isize_field, 3) ...): the UNCHANGED time loop (link-time RHS + the reference's DAXPY / DSCAL, which the library
            # completes to one fused substep: csrc/deferred.cpp) and the six-line patch of time.f90, ms per substep by the driver's own clock
            exe = os.path.join(ROOT, "tlab_amd", "fortran", "_build_rk", "test_rk_driver")
            if os.path.exists(exe):
                d = None
                torch.cuda.empty_cache()
                try:
                    import subprocess
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fortran_host.py"), "--grid", str(nx), str(ny), str(nz), "--steps", "6", "--warmup", "1",
                                        "--routes", "unchanged,fused"], capture_output=True, text=True, timeout=900)
                    fl = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    out["fortran_host"] = json.loads(fl[-1]) if fl else {"error": (r.stdout + r.stderr)[-800:]}
                except Exception as e:       # noqa: BLE001
                    out["fortran_host"] = {"error": repr(e)}
            else:
                out["fortran_host"] = {"error": "tlab_amd/fortran/_build_rk/test_rk_driver not built (needs the reference's module files: build container only)"}
        if args.cpu_sample > 0 and world == 1 and args.loopback <= 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.nscal)
            # ... and the host used the way the reference uses it (one rank per core): independent instances on disjoint core ranges.  The larger of
            # the two is `cpu_baseline`; the other one stays in the record.
            multi = None
            try:      # only where the job is granted more CPUs than one instance uses (on this pool it is not: 16 granted, 16 used)
                if cpu_quota()[0] >= 2 * int(out["cpu_baseline"].get("cores", 16)):
                    multi = cpu_baseline_instances(args.cpu_sample, args.nscal, int(out["cpu_baseline"].get("cores", 16)))
            except Exception as e:       # noqa: BLE001
                multi = {"error": repr(e)}
            if multi and "value" in multi and multi["value"] > out["cpu_baseline"]["value"]:
                out["cpu_baseline_single_instance"] = out["cpu_baseline"]
                out["cpu_baseline"] = multi
            elif multi:
                out["cpu_baseline_instances"] = multi
            if args.cpu_sample_large > args.cpu_sample and (os.cpu_count() or 1) >= args.cpu_large_min_cores:
                # the benchmark's own size on the host cores, when the host is big enough to finish it in about a minute
                large = cpu_baseline(args.cpu_sample_large, args.nscal)
                if isinstance(large, dict) and large.get("value"):      # like for like: the benchmark's own size is the headline baseline when it ran
                    out["cpu_baseline_small_sample"] = out["cpu_baseline"]
                    out["cpu_baseline"] = large
                else:
                    out["cpu_baseline_large"] = large
        if "cpu_baseline" in out:
            # ... and the reference's OWN compiled routines composed into the same substep (oracle/tlab_ref_rhs.py: every derivative, Burgers operator and
            # per-mode solve is the reference's Fortran built from /root/reference by oracle/Makefile; FFTW is not in the image, so the transforms and the
            # loop over the Fourier modes are numpy / Python), serial, on a 128^3 sample: kind "reference" next to the parallel "port" above
            try:
                out["cpu_baseline_reference"] = cpu_baseline_reference(128, args.nscal)
            except Exception as e:       # noqa: BLE001
                out["cpu_baseline_reference"] = {"error": repr(e)}
        print(json.dumps(out))
    # the native slab driver owns an RCCL communicator of its own: release it while every rank is still here, not during interpreter teardown
    try:
        if hasattr(d, "close"):
            torch.cuda.synchronize()
            d.close()
    except Exception as e:       # noqa: BLE001  (the measurement is already printed)
        print("bench.py: closing the slab driver: %r" % (e,), file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

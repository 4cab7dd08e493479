"""Multi-PROCESS run of the z-slab algorithm on the GPU box: P ranks (processes) share the one GPU, torch.distributed runs on gloo
with host-staged payloads (tlab_amd/parallel.py DistComm.stage_host), every arithmetic operation is the HIP library.  Checks the
per-process code path the 8-GPU job uses (rank-local plans, neighbour pairing, uneven kx-pencils) against the single-domain step."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, port, script_args, env, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


# (driver, zmode, nz, bcs, nx) per world size -- ONE torch.distributed.run launch per world size (round 4 started one per case: twelve interpreter
# start-ups of 4-9 s each inside the driver's time limit); every case prints its own DIST_CHECK line and all of them are asserted
GLOO_CASES = {
    2: [("python", "halo", 128, "noslip", 32), ("python", "transpose", 64, "noslip", 32), ("native", "auto", 128, "noslip", 32), ("native", "auto", 128, "noslip", 128)],
    3: [("python", "halo", 192, "freeslip", 48), ("native", "auto", 192, "freeslip", 48), ("native", "auto", 192, "freeslip", 128)],
}


@pytest.mark.parametrize("world", [2, 3])
def test_multiprocess_slab_drivers(world):
    """P processes share the one GPU (gloo, host-staged payloads).  Python driver (tlab_amd/parallel.py::SlabDns): halo and transposition schemes.  Native
    C++ driver (tlab_slab_dns_*) with the five transport entry points supplied by the caller (ctypes callbacks over gloo, tlab_amd/slab.py::dist_transport):
    rank-local plans, ring pairing with 2 and 3 ranks, uneven kx-pencils in two halves, monitors -- against the single-domain step each rank computes
    redundantly.  nx = 128: the repack passes are folded into the library's own x-transforms and (no-slip) v is finished by the inverse transform of dp^/dy."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, TLAB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cases = GLOO_CASES[world]
    spec = ";".join("%s:%s:%d:%s:%d" % c for c in cases)
    out = _launch(world, 29540 + world, [os.path.join(ROOT, "tools", "dist_check.py"), "--cases", spec], env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DIST_CHECK")]
    assert len(lines) == len(cases), out.stdout[-3000:]
    for (drv, zm, nz, bcs, nx), ln in zip(cases, lines):
        assert ("driver=%s world=%d" % (drv, world)) in ln and ("nx=%d nz=%d bcs=%s" % (nx, nz, bcs)) in ln and ln.endswith(" OK"), ln
        if drv == "python":
            assert "zmode=%s" % zm in ln, ln
        else:
            assert ("fused_x=1" if nx == 128 else "fused_x=0") in ln, ln


def test_native_slab_driver_over_rccl_world_size_one():
    """The product transport (grouped ncclSend / ncclRecv on the communication stream of libtlab_amd_comm.so, events against the compute stream,
    ncclAllReduce of the monitors) under the C++ slab driver.  One GPU here, so one rank that is its own ring neighbour and all-to-all peer: every
    call, the ticket / event logic and the stream ordering are exercised, traffic between different GPUs is not."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TLAB_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "tools", "dist_check.py"), "--cases", "native:auto:64:noslip:32;python:transpose:128:noslip:32"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DIST_CHECK")]
    assert len(lines) == 2, out.stdout[-2000:]
    assert "driver=native world=1" in lines[0] and "backend=nccl" in lines[0] and "rccl_ranks=1" in lines[0] and lines[0].endswith(" OK"), lines[0]
    # ... and the Python driver's DistComm on torch's NCCL group with device buffers (the fall-back of bench.py --gpus N), transposition scheme
    assert "driver=python world=1 zmode=transpose backend=nccl" in lines[1] and lines[1].endswith(" OK"), lines[1]


def test_native_slab_driver_over_rccl_with_a_gloo_start_up():
    """bench.py's default start-up for N > 1: the torch.distributed group is gloo -- it carries the ncclUniqueId and the control
    reductions -- so that the RCCL inside libtlab_amd_comm.so is the only RCCL in the process (torch's bundled copy is never initialised)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TLAB_DIST_BOOTSTRAP="gloo")
    env.pop("TLAB_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29563", os.path.join(ROOT, "tools", "dist_check.py"), "--driver", "native", "--nz", "64"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "DIST_CHECK driver=native world=1" in out.stdout and "backend=nccl bootstrap=gloo" in out.stdout and " OK" in out.stdout, out.stdout[-2000:]


def test_rccl_operations_of_the_slab_driver():
    """DistComm on the RCCL backend with device buffers (what `bench.py --gpus N` uses).  The test box has one GPU, so one rank:
    every peer is the rank itself, but the calls, the grouped send/recv and the stream ordering are RCCL's.  Followed by the
    slab substep on that backend against the single-domain substep."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TLAB_DIST_BACKEND", None)
    for script, mark, extra in (("rccl_check.py", "RCCL_CHECK", []),):      # (its slab substep on that backend: second case of the test above)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29557", os.path.join(ROOT, "tools", script)] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        assert mark in out.stdout and " OK" in out.stdout, out.stdout[-2000:]
        assert "backend=nccl" in out.stdout or script == "rccl_check.py"


@pytest.mark.parametrize("world,n", [(2, 128)])
def test_bench_contract_with_two_ranks(world, n):
    """`bench.py --gpus N` exactly as the driver launches it for N > 1 -- torch.distributed.run, one rank per process, barrier + MAX over ranks,
    ONE JSON line from rank 0 -- here with the ranks sharing the one GPU of the test box (TLAB_DIST_BACKEND=gloo: the native slab driver over the
    caller-supplied transport; with RCCL the same code path needs one GPU per rank)."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, TLAB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29593", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--box", str(n)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "strong" and rec["higher_is_better"] is True
    assert rec["value"] > 0 and abs(rec["value"] - n ** 3 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]
    assert rec["config"]["fields_finite"] is True and "z-slabs 1x%d" % world in rec["config"]["parallelism"]

"""GPU test of the native transposition layer (libtlab_amd_comm.so: TLabMPI_Trp_* over RCCL behind the C ABI, include/tlab_amd_comm.h).
tools/native_comm_check.py runs in a process of its own without torch: P ranks looped back through tlab_trp_pack / tlab_trp_unpack
(bit-exact against the closed form of base/tlab_mpi_transpose.f90:232-256, :301-325; I and K, real and complex, P = 1..8) and the RCCL path
(communicator, tlab_trp_exec / start + wait, all-reduce) on the ranks available -- one on the single-GPU box."""
import os
import subprocess
import sys

import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_native_transpositions_loopback_and_rccl(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "native_comm_check.py"), "--idfile", str(tmp_path / "id")],
                       capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr[-3000:])
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "native comm ok" in r.stdout

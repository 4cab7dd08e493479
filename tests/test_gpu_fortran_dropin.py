"""GPU test of the Fortran side of the boundary: tlab_amd/fortran/test_dropin.f90 calls the drop-in modules
(OPR_Partial / OPR_Burgers signatures of the reference, ISO_C_BINDING underneath) on device memory and compares with the
reference's own CPU modules (oracle/_ref objects) in the same executable."""
import os
import subprocess
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "tlab_amd", "fortran", "_build", "test_dropin")


def test_fortran_dropin_side_by_side():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(EXE):
        pytest.skip("tlab_amd/fortran/_build/test_dropin not built (needs oracle/_ref, i.e. the build container)")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin ok" in r.stdout

"""grid and field files (base/tlab_grid.f90, base/io_fields.f90) -- host-side byte formats.
Fixtures in tests/golden/io_* were written by the reference's own routines (make_golden_io.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_io import NX, NY, NZ, inputs  # noqa: E402
from tlab_amd import io as tio  # noqa: E402
from tlab_amd.lib import TlabError  # noqa: E402

G = os.path.join(HERE, "golden")


def test_reads_reference_written_files():
    x, y, z, f = inputs()
    xr, yr, zr, sc = tio.grid_read(os.path.join(G, "io_grid"), (NX, NY, NZ))
    assert np.array_equal(xr, x) and np.array_equal(yr, y) and np.array_equal(zr, z)
    assert np.array_equal(sc, [x[-1] - x[0], y[-1] - y[0], z[-1] - z[0]])
    flow, nt, params = tio.io_read_fields(os.path.join(G, "io_flow"), NX, NY, NZ, 3)
    assert nt == 1234 and np.array_equal(params, [0.5, 2.0e-4, 3.0, 1.0])
    assert all(np.array_equal(a, b) for a, b in zip(flow, f[:3]))
    scal, nt, params = tio.io_read_fields(os.path.join(G, "io_scal"), NX, NY, NZ, 1)
    assert nt == 1234 and params.size == 0 and np.array_equal(scal[0], f[3])
    only2, _, _ = tio.io_read_fields(os.path.join(G, "io_flow"), NX, NY, NZ, 3, iread=2)
    assert len(only2) == 1 and np.array_equal(only2[0], f[1])


def test_writes_byte_identical_files(tmp_path):
    x, y, z, f = inputs()
    tio.grid_write(str(tmp_path / "grid"), x, y, z)
    assert open(tmp_path / "grid", "rb").read() == open(os.path.join(G, "io_grid"), "rb").read()
    tio.io_write_fields(str(tmp_path / "flow"), NX, NY, NZ, 1234, f[:3], [0.5, 2.0e-4, 3.0, 1.0])
    tio.io_write_fields(str(tmp_path / "scal"), NX, NY, NZ, 1234, f[3:], [])
    for name, ref in (("flow.1", "io_flow.1"), ("flow.2", "io_flow.2"), ("flow.3", "io_flow.3"), ("scal.1", "io_scal.1")):
        assert open(tmp_path / name, "rb").read() == open(os.path.join(G, ref), "rb").read(), name


def test_error_paths(tmp_path):
    with pytest.raises(TlabError):                       # DNS_ERROR_DIMGRID
        tio.grid_read(os.path.join(G, "io_grid"), (NX, NY, NZ + 1))
    with pytest.raises(TlabError):                       # IO_READ_HEADER. Grid size mismatch.
        tio.io_read_fields(os.path.join(G, "io_flow"), NX + 1, NY, NZ, 1)
    bad = tmp_path / "bad.1"
    np.array([23, NX, NY, NZ, 0], dtype="<i4").tofile(bad)
    with pytest.raises(TlabError):                       # IO_READ_HEADER. Header format incorrect.
        tio.io_read_header(str(bad))


def test_live_against_reference_build(tmp_path):
    from oracle import ref_lib as R
    if not R.available():
        pytest.skip("oracle/_ref not built here")
    x, y, z, f = inputs()
    R.init(NX, NY, NZ)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        tio.io_write_fields("own", NX, NY, NZ, 5, f[:2], [1.5])
        got, p = R.io_read_fields("own", NX, NY, NZ, 5, 2, 1)           # the reference reads what we wrote
        assert p[0] == 1.5 and all(np.array_equal(a, b) for a, b in zip(got, f[:2]))
        tio.grid_write("g", x, y, z)
        xr, yr, zr, sc = R.grid_read("g", NX, NY, NZ)
        assert np.array_equal(yr, y) and sc[2] == z[-1] - z[0]
    finally:
        os.chdir(cwd)


@pytest.mark.gpu
def test_device_driver_round_trip_through_reference_written_files(tmp_path):
    """n4 on the device path: Dns.load_fields reads the restart files the reference's IO_Write_Fields wrote (tests/golden/io_*) into HBM, a
    Runge-Kutta step runs on them, Dns.save_fields writes files the reference's tools read back; an untouched load -> save round trip is
    byte-identical to the reference's files (the Fortran host takes the same route through IO_Fields_AMD, tests/test_gpu_fortran_dropin.py)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tlab_amd as T
    from tlab_amd.dns import Dns
    T.init(0)
    x, y, z, f = inputs()
    xr, yr, zr, _ = tio.grid_read(os.path.join(G, "io_grid"), (NX, NY, NZ))
    d = Dns(xr, yr, zr, nscal=1, visc=2.0e-4, schmidt=(1.0,), yuniform=False)
    nt, params = d.load_fields(os.path.join(G, "io_flow"), os.path.join(G, "io_scal"))
    assert nt == 1234
    for t, a in zip(d.q + d.s, f):
        assert np.array_equal(t.cpu().numpy(), a)
    d.save_fields(str(tmp_path / "flow"), None, nt=1234, params=[0.5, 2.0e-4, 3.0, 1.0])
    d.save_fields(None, str(tmp_path / "scal"), nt=1234, params=[])
    for name, ref in (("flow.1", "io_flow.1"), ("flow.2", "io_flow.2"), ("flow.3", "io_flow.3"), ("scal.1", "io_scal.1")):
        assert open(tmp_path / name, "rb").read() == open(os.path.join(G, ref), "rb").read(), name
    d.TIME_RUNGEKUTTA(1e-4)                      # the loaded state is a usable state
    d.save_fields(str(tmp_path / "flow2"), str(tmp_path / "scal2"), nt=1235, params=[0.5001, 2.0e-4])
    back, nt2, p2 = tio.io_read_fields(str(tmp_path / "flow2"), NX, NY, NZ, 3)
    assert nt2 == 1235 and all(np.isfinite(a).all() for a in back) and np.array_equal(back[0], d.q[0].cpu().numpy())

#!/usr/bin/env python3
"""Restart-file fixtures written BY THE REFERENCE's own routines (TLab_Grid_Write base/tlab_grid.f90:72, IO_Write_Fields
base/io_fields.f90:346, compiled in place into oracle/_ref): tests/golden/io_grid, io_flow.1..3 (with header parameters), io_scal.1
(without).  Run where /root/reference exists:  make -C oracle && python3 tests/golden/make_golden_io.py"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_lib as R  # noqa: E402

NX, NY, NZ = 12, 10, 8


def inputs():
    x = np.arange(NX) / NX
    y = 0.5 * (1 + np.tanh(2 * (2 * np.arange(NY) / (NY - 1) - 1)) / np.tanh(2))
    z = np.arange(NZ) / NZ * 2.0
    rng = np.random.default_rng(20250509)
    fields = [rng.uniform(-1, 1, NX * NY * NZ) for _ in range(4)]
    return x, y, z, fields


if __name__ == "__main__":
    if not R.available():
        sys.exit("oracle/_ref/libtlab_ref.so missing")
    R.init(NX, NY, NZ)
    x, y, z, f = inputs()
    os.chdir(HERE)
    R.grid_write("io_grid", x, y, z)
    R.io_write_fields("io_flow", NX, NY, NZ, 1234, f[:3], [0.5, 2.0e-4, 3.0, 1.0])      # (rtime, visc, froude-like, ...) as dns.x would
    R.io_write_fields("io_scal", NX, NY, NZ, 1234, f[3:], [])
    print("wrote io_grid, io_flow.1-3, io_scal.1")

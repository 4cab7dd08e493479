"""Restart-file formats of the reference, host side -- what lets a run on the device path start from, and be
post-processed by, the reference's own tools.

  grid file     TLab_Grid_Read / TLab_Grid_Write   base/tlab_grid.f90:26,72
                Fortran sequential unformatted: records (nx, ny, nz) int32 | (scale_x, scale_y, scale_z) real64 | x | y | z
  field files   IO_Read_Fields / IO_Write_Fields   base/io_fields.f90:150,346 (serial branch, io_datatype = double)
                one stream-access file per field, "<name>.<ifield>": header = int32 (offset, nx, ny, nz, nt) [+ real64 params],
                offset = 5*4 + 8*len(params) (IO_WRITE_HEADER :578-597); then nx*ny*nz real64, x fastest, at byte `offset`
                (files are opened access='stream', include/dns_open_file.h: no record markers).
Byte-identical to what the reference's routines write (tests/test_io_formats.py: fixtures made by the reference + live comparison
against oracle/_ref where it is present)."""
import numpy as np

from .lib import TlabError


def _record(f, dtype, count):
    n = np.fromfile(f, dtype="<i4", count=1)
    if n.size != 1 or n[0] != np.dtype(dtype).itemsize * count:
        raise TlabError("grid file: unexpected record length")
    a = np.fromfile(f, dtype=dtype, count=count)
    m = np.fromfile(f, dtype="<i4", count=1)
    if m.size != 1 or m[0] != n[0]:
        raise TlabError("grid file: broken record")
    return a


def grid_read(name, sizes=None):
    """Returns (x, y, z, scales).  sizes: optional (nx, ny, nz) to check against (DNS_ERROR_DIMGRID in the reference)."""
    with open(name, "rb") as f:
        n = _record(f, "<i4", 3)
        if sizes is not None and tuple(int(v) for v in n) != tuple(int(v) for v in sizes):
            raise TlabError("grid file: dimensions (%d,%d,%d) unmatched" % tuple(n))
        scales = _record(f, "<f8", 3)
        x, y, z = (_record(f, "<f8", int(k)) for k in n)
    return x, y, z, scales


def grid_write(name, x, y, z, scales=None):
    x, y, z = (np.ascontiguousarray(a, dtype="<f8") for a in (x, y, z))
    if scales is None:
        scales = [a[-1] - a[0] for a in (x, y, z)]

    def rec(f, a):
        m = np.array([a.nbytes], dtype="<i4")
        m.tofile(f); a.tofile(f); m.tofile(f)
    with open(name, "wb") as f:
        rec(f, np.array([x.size, y.size, z.size], dtype="<i4"))
        rec(f, np.asarray(scales, dtype="<f8"))
        rec(f, x); rec(f, y); rec(f, z)


def io_write_fields(fname, nx, ny, nz, nt, fields, params=()):
    """fields: sequence of flat arrays of nx*ny*nz (x fastest).  Writes fname.1, fname.2, ...; params go into every header."""
    params = np.asarray(params, dtype="<f8")
    offset = 5 * 4 + 8 * params.size
    for i, a in enumerate(fields):
        a = np.ascontiguousarray(a, dtype="<f8").reshape(-1)
        if a.size != nx * ny * nz:
            raise TlabError("io_write_fields: field %d has %d values, expected %d" % (i + 1, a.size, nx * ny * nz))
        with open("%s.%d" % (fname, i + 1), "wb") as f:
            np.array([offset, nx, ny, nz, nt], dtype="<i4").tofile(f)
            params.tofile(f)
            a.tofile(f)


def io_read_header(name):
    with open(name, "rb") as f:
        h = np.fromfile(f, dtype="<i4", count=5)
        if h.size != 5:
            raise TlabError("field file: truncated header")
        isize = int(h[0]) - 5 * 4
        if isize < 0 or isize % 8:
            raise TlabError("IO_READ_HEADER. Header format incorrect.")
        params = np.fromfile(f, dtype="<f8", count=isize // 8)
    return int(h[0]), (int(h[1]), int(h[2]), int(h[3])), int(h[4]), params


def io_read_fields(fname, nx, ny, nz, nfield, iread=0):
    """Returns (fields, nt, params).  iread = 0 reads all nfield files, otherwise only field iread (io_fields.f90:152)."""
    out, nt, params = [], None, None
    for i in range(1, nfield + 1):
        if iread not in (0, i):
            continue
        name = "%s.%d" % (fname, i)
        offset, dims, nt, params = io_read_header(name)
        if dims != (nx, ny, nz):
            raise TlabError("IO_READ_HEADER. Grid size mismatch.")
        a = np.fromfile(name, dtype="<f8", count=nx * ny * nz, offset=offset)
        if a.size != nx * ny * nz:
            raise TlabError("field file %s: truncated" % name)
        out.append(a)
    return out, nt, params

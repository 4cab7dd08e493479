#!/usr/bin/env python3
"""Dependency scanner for building the reference's hot-path Fortran sources where they lie.

TEST INFRASTRUCTURE ONLY.  Scans `use <module>` statements under /root/reference/src and prints the
topologically ordered list of source files needed to compile the requested root files.  It never
copies a reference source; the Makefile next to it compiles them in place into oracle/_ref/.
"""
import os, re, sys

SRC = os.environ.get("TLAB_REFERENCE", "/root/reference") + "/src"
use_re = re.compile(r"^\s*use\s+(?:,\s*intrinsic\s*::\s*)?([a-z0-9_]+)", re.I)
mod_re = re.compile(r"^\s*module\s+([a-z0-9_]+)\s*$", re.I)
INTRINSIC = {"iso_c_binding", "iso_fortran_env", "omp_lib", "mpi_f08", "mpi", "netcdf", "ieee_arithmetic"}


def scan(defines=()):
    provides, uses = {}, {}
    for d, _, fs in os.walk(SRC):
        if "/valid" in d or "/tools" in d or "/old" in d:
            continue
        for f in fs:
            if not f.endswith(".f90"):
                continue
            p = os.path.join(d, f)
            us = set()
            # honour simple #ifdef USE_MPI / #else / #endif blocks (we never define USE_MPI etc.)
            stack = []
            for line in open(p, errors="replace"):
                s = line.strip()
                if s.startswith("#ifdef"):
                    stack.append(s.split()[1] in defines); continue
                if s.startswith("#ifndef"):
                    stack.append(s.split()[1] not in defines); continue
                if s.startswith("#if "):
                    stack.append(True); continue
                if s.startswith("#else"):
                    if stack: stack[-1] = not stack[-1]
                    continue
                if s.startswith("#endif"):
                    if stack: stack.pop()
                    continue
                if stack and not all(stack):
                    continue
                m = mod_re.match(line)
                if m and m.group(1).lower() != "procedure":
                    provides[m.group(1).lower()] = p
                m = use_re.match(line)
                if m:
                    us.add(m.group(1).lower())
            uses[p] = us
    return provides, uses


def closure(roots, extra=()):
    provides, uses = scan()
    order, seen = [], set()

    def visit(p):
        if p in seen:
            return
        seen.add(p)
        for m in sorted(uses[p]):
            if m in INTRINSIC:
                continue
            if m not in provides:
                print("WARNING: module %s (used by %s) not found" % (m, p), file=sys.stderr)
                continue
            visit(provides[m])
        order.append(p)

    for r in list(extra) + list(roots):
        visit(os.path.join(SRC, r))
    return order


if __name__ == "__main__":
    for p in closure(sys.argv[1:]):
        print(p)

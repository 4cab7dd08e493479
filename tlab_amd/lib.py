"""ctypes binding of libtlab_amd.so (the C ABI declared in include/tlab_amd.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TlabError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libtlab_amd.so")


c_int, c_dbl, c_vp, c_sz = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); must list every entry point of include/tlab_amd.h (tests/test_capi_symbols.py checks)
SIGNATURES = {
    "tlab_init": (c_int, [c_int]),
    "tlab_device_count": (c_int, []),
    "tlab_finalize": (c_int, []),
    "tlab_last_error": (ctypes.c_char_p, []),
    "tlab_set_stream": (c_int, [c_vp]),
    "tlab_sync": (c_int, []),
    "tlab_malloc": (c_int, [ctypes.POINTER(c_vp), c_sz]),
    "tlab_free": (c_int, [c_vp]),
    "tlab_memcpy_h2d": (c_int, [c_vp, c_vp, c_sz]),
    "tlab_memcpy_d2h": (c_int, [c_vp, c_vp, c_sz]),
    "tlab_fdm_plan_create": (c_int, [ctypes.POINTER(c_vp), c_int, _dp, c_int, c_int, c_int, c_int, c_dbl]),
    "tlab_fdm_plan_create_from_arrays": (c_int, [ctypes.POINTER(c_vp), c_int, c_int, c_int, c_int, c_int, _dp, _dp, c_int, c_int, _dp, _dp]),
    "tlab_fdm_plan_set_aux": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tlab_fdm_plan_set_scheme": (c_int, [c_vp, c_int, c_int]),
    "tlab_fdm_plan_set_stagger": (c_int, [c_vp, c_int]),
    "tlab_fdm_plan_destroy": (c_int, [c_vp]),
    "tlab_fdm_plan_get": (c_int, [c_vp, c_int, _dp, c_int]),
    "tlab_fdm_plan_info": (c_int, [c_vp, c_int]),
    "tlab_opr_partial": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tlab_opr_burgers": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_int]),
    "tlab_poisson_plan_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_int, c_int, c_int]),
    "tlab_poisson_plan_destroy": (c_int, [c_vp]),
    "tlab_opr_poisson": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tlab_poisson_plan_create_slab": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int]),
    "tlab_poisson_plan_create_pencil": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int]),
    "tlab_pencil_repack": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int]),
    "tlab_pencil_repack_blocks": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_longlong), c_int]),
    "tlab_poisson_set_wall_planes": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "tlab_poisson_fft_x": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_poisson_fft_x_packed": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tlab_poisson_fft_x_packed_final": (c_int, [c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_int, c_vp, c_vp]),
    "tlab_poisson_fft_z": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_poisson_ode": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "tlab_poisson_plan_create_direct": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "tlab_poisson_direct_ode": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_poisson_set_exact": (c_int, [c_int]),
    "tlab_poisson_plan_create_direct_decomposed": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tlab_opr_helmholtz": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tlab_pw_add3": (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_longlong]),
    "tlab_pw_axpy3": (c_int, [c_vp] * 9 + [c_dbl, ctypes.c_longlong]),
    "tlab_pw_sum3": (c_int, [c_vp, c_vp, c_vp, ctypes.c_longlong]),
    "tlab_pw_sub3": (c_int, [c_vp] * 6 + [ctypes.c_longlong]),
    "tlab_pw_rk_update": (c_int, [c_vp, c_vp, c_dbl, c_dbl, c_int, ctypes.c_longlong]),
    "tlab_pw_fill": (c_int, [c_vp, c_dbl, ctypes.c_longlong]),
    "tlab_pw_scale": (c_int, [c_vp, c_dbl, ctypes.c_longlong]),
    "tlab_pw_final_update": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_int, c_int, c_int]),
    "tlab_pw_get_wall_planes": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int]),
    "tlab_pw_fill_wall_planes": (c_int, [c_vp, c_dbl, c_dbl, c_int, c_int, c_int]),
    "tlab_dns_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_dbl, _dp]),
    "tlab_dns_destroy": (c_int, [c_vp]),
    "tlab_dns_set_fusion": (c_int, [c_vp, c_int]),
    "tlab_dns_begin_step": (c_int, [c_vp]),
    "tlab_deferred_enable": (c_int, [c_int]),
    "tlab_deferred_rhs": (c_int, [c_vp, c_dbl, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "tlab_deferred_axpy": (c_int, [ctypes.c_longlong, c_dbl, c_vp, c_vp]),
    "tlab_deferred_scal": (c_int, [ctypes.c_longlong, c_dbl, c_vp]),
    "tlab_deferred_zero": (c_int, [c_vp, ctypes.c_longlong]),
    "tlab_deferred_flush": (c_int, []),
    "tlab_deferred_slab_rhs": (c_int, [c_vp, c_dbl]),
    "tlab_deferred_pencil_rhs": (c_int, [c_vp, c_dbl]),
    "tlab_deferred_stats": (c_int, [ctypes.POINTER(ctypes.c_longlong)]),
    "tlab_dns_place_arrays": (c_int, [c_vp, c_int, c_vp, c_vp, c_dbl, c_int, ctypes.c_uint, ctypes.POINTER(c_int), _dp]),
    "tlab_dns_place_blocks": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                              ctypes.c_longlong, c_dbl, c_int, ctypes.c_uint, ctypes.POINTER(c_int), _dp]),
    "tlab_dns_set_slab": (c_int, [c_vp, c_int]),
    "tlab_dns_set_anelastic": (c_int, [c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    "tlab_dns_set_remove_divergence": (c_int, [c_vp, c_int]),
    "tlab_dns_set_surface_bcs": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    "tlab_opr_burgers_set_anelastic": (c_int, [c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    "tlab_filter_create": (c_int, [ctypes.POINTER(c_vp), c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_dbl)]),
    "tlab_filter_destroy": (c_int, [c_vp]),
    "tlab_opr_filter_1d": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "tlab_opr_burgers_set_dealiasing": (c_int, [c_int, c_vp]),
    "tlab_opr_filter": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(c_int), c_vp, c_vp]),
    "tlab_dns_set_pressure_filter": (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_int)]),
    "tlab_time_courant": (c_int, [c_vp, ctypes.POINTER(c_vp), c_dbl, c_dbl, _dp, _dp]),
    "tlab_fi_invariant_p": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tlab_minmax": (c_int, [c_vp, c_vp, c_int, c_int, c_int, _dp, _dp]),
    "tlab_opr_burgers_add": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tlab_opr_burgers_add_n": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, _dp, ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp), c_vp, c_vp, c_int]),
    "tlab_opr_partial_add": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_dbl, c_vp, c_int, c_vp, c_vp]),
    "tlab_opr_gradient_final": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_vp]),
    "tlab_zslab_burgers_z_n": (c_int, [c_vp, c_int, c_int, c_int, c_int, _dp, ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp), c_int]),
    "tlab_zslab_gradient_final_z": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "tlab_zslab_plan_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_int, c_int, c_int]),
    "tlab_zslab_plan_destroy": (c_int, [c_vp]),
    "tlab_zslab_partial_z": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "tlab_zslab_burgers_z": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "tlab_dns_set_bcs": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "tlab_boundary_bcs_neumann_y": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tlab_pw_set_wall_planes": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int]),
    "tlab_rhs_global_incompressible_1": (c_int, [c_vp, c_dbl, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "tlab_time_substep_incompressible_explicit": (c_int, [c_vp, c_dbl, c_dbl, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "tlab_slab_transport_loopback": (c_int, [c_vp, c_int]),
    "tlab_slab_dns_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_dbl, _dp, c_vp]),
    "tlab_slab_dns_destroy": (c_int, [c_vp]),
    "tlab_slab_dns_bind": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "tlab_slab_dns_info": (ctypes.c_longlong, [c_vp, c_int]),
    "tlab_slab_dns_set_bcs": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "tlab_slab_dns_begin_step": (c_int, [c_vp]),
    "tlab_slab_dns_set_remove_divergence": (c_int, [c_vp, c_int]),
    "tlab_slab_dns_set_surface_bcs": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    "tlab_pencil_transport_loopback": (c_int, [c_vp, c_int, c_int]),
    "tlab_pencil_dns_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_dbl, ctypes.POINTER(c_dbl)]),
    "tlab_pencil_dns_destroy": (c_int, [c_vp]),
    "tlab_pencil_dns_bind": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "tlab_pencil_dns_info": (ctypes.c_longlong, [c_vp, c_int]),
    "tlab_pencil_dns_set_bcs": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "tlab_pencil_dns_begin_step": (c_int, [c_vp]),
    "tlab_pencil_dns_rhs": (c_int, [c_vp, c_dbl]),
    "tlab_pencil_dns_trace": (c_int, [c_vp, c_int, ctypes.c_char_p, c_int]),
    "tlab_pencil_dns_substep": (c_int, [c_vp, c_dbl, c_dbl, c_int]),
    "tlab_slab_dns_rhs": (c_int, [c_vp, c_dbl]),
    "tlab_slab_dns_substep": (c_int, [c_vp, c_dbl, c_dbl, c_int]),
    "tlab_slab_dns_time_courant": (c_int, [c_vp, c_dbl, c_dbl, _dp, _dp]),
    "tlab_slab_dns_dilatation_bounds": (c_int, [c_vp, _dp, _dp]),
    "tlab_transpose": (c_int, [c_vp, c_int, c_int, c_vp]),
    "tlab_last_kernel_path": (c_int, []),
    "tlab_force_kernel_path": (c_int, [c_int]),
    "tlab_set_tuning": (c_int, [c_int, c_int]),
    "tlab_profile_enable": (c_int, [c_int]),
    "tlab_profile_filter": (c_int, [ctypes.c_char_p]),
    "tlab_profile_reset": (c_int, []),
    "tlab_profile_report": (c_int, [ctypes.c_char_p, c_int]),
    "tlab_debug_host_chunked_solve": (c_int, [c_vp, c_int, c_int, c_int, _dp]),
    "tlab_debug_pack_map": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tlab_debug_int1_tables": (c_int, [c_vp, c_int, c_int, _dp, _dp, _dp, _dp, _dp]),
    "tlab_debug_int1_solve": (c_int, [c_vp, c_int, c_int, c_int, _dp, _dp, _dp, _dp, _dp]),
}


def load():
    """Load the HIP library; fails loudly when it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise TlabError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C tlab_amd/csrc` (there is no CPU fallback)" % path)
    L = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)       # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(code, what=""):
    if code != 0:
        msg = load().tlab_last_error()
        raise TlabError("%s failed (%d): %s" % (what, code, msg.decode() if msg else ""))

!########################################################################
! Side-by-side test of the Fortran drop-in: the reference's own FDM_CreatePlan builds the host plans, then the
! reference's CPU OPR_Partial_{X,Y,Z} (module OPR_Partial from oracle/_ref) and the device drop-in (compiled here as
! OPR_Partial_AMD) run on the same field; OPR_Burgers_{X,Y,Z} of the drop-in is checked against nu*P2 - u*P1 of the CPU
! module (the construction of src/valid/burgers/vburgers.f90:78-151).  Prints the relative errors; exit code 0 iff all <= 1e-12.
!########################################################################
program test_dropin
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi, BCS_NN, BCS_DD
    use TLab_Arrays, only: wrk1d, wrk2d, wrk3d
    use TLab_OpenMP, only: TLab_OMP_numThreads
    use TLab_Grid, only: grid_dt
    use FDM, only: fdm_dt, FDM_CreatePlan
    use FDM_Derivative, only: FDM_COM6_JACOBIAN, FDM_COM6_JACOBIAN_HYPER, FDM_COM6_DIRECT
    use OPR_Partial, only: CPU_Partial_X => OPR_Partial_X, CPU_Partial_Y => OPR_Partial_Y, CPU_Partial_Z => OPR_Partial_Z, &
                           OPR_P1, OPR_P2, OPR_P2_P1
    use OPR_Partial_AMD, only: GPU_Partial_X => OPR_Partial_X, GPU_Partial_Y => OPR_Partial_Y, GPU_Partial_Z => OPR_Partial_Z
    use OPR_Burgers_AMD
    use OPR_Elliptic_AMD
    use TLab_AMD_C
    implicit none

    integer(wi), parameter :: nx = 256, ny = 96, nz = 64
    integer(wi), parameter :: n = nx*ny*nz
    type(grid_dt) :: gr(3)
    type(fdm_dt), target :: g(3), fdm_loc
    real(wp), allocatable, target :: u(:), v(:), r_cpu(:), t_cpu(:), r_gpu(:), b_ref(:)
    real(wp), pointer :: d_u(:), d_v(:), d_r(:), d_t(:), d_t1(:), d_t2(:), d_hb(:), d_ht(:)
    type(c_ptr) :: p_u, p_v, p_r, p_t, p_t1, p_t2, p_hb, p_ht
    real(wp), allocatable, target :: phi(:), dphidy(:), f(:), w1(:), w2(:), hb(:), ht(:)
    integer(wi) :: ntxc
    integer(wi) :: bcs(2, 2), i, j, k, ig, sizes(3)
    integer(c_int) :: rc
    real(wp) :: err, worst, visc
    real(wp), parameter :: pi = 3.14159265358979323846_wp

    TLab_OMP_numThreads = 1
    allocate (wrk1d(max(nx, ny, nz), 20), wrk2d(max(nx*ny, nx*nz, ny*nz), 6), wrk3d((nx + 2)*ny*nz))
    sizes = [nx, ny, nz]
    do ig = 1, 3
        gr(ig)%name = 'xyz'(ig:ig); gr(ig)%size = sizes(ig); gr(ig)%periodic = (ig /= 2)
        allocate (gr(ig)%nodes(sizes(ig)))
        if (ig == 2) then       ! tanh-stretched, non-periodic
            gr(ig)%nodes = [(0.5_wp*(1.0_wp + tanh(2.0_wp*(2.0_wp*real(i - 1, wp)/real(ny - 1, wp) - 1.0_wp))/tanh(2.0_wp)), i=1, ny)]
        else
            gr(ig)%nodes = [(real(i - 1, wp)/real(sizes(ig), wp), i=1, sizes(ig))]
        end if
        gr(ig)%scale = gr(ig)%nodes(sizes(ig)) - gr(ig)%nodes(1)
        g(ig)%name = gr(ig)%name; g(ig)%periodic = gr(ig)%periodic; g(ig)%uniform = (ig /= 2)
        g(ig)%der1%mode_fdm = FDM_COM6_JACOBIAN; g(ig)%der2%mode_fdm = FDM_COM6_JACOBIAN_HYPER
        call FDM_CreatePlan(gr(ig), g(ig))                   ! the reference's own, unchanged
    end do

    allocate (u(n), v(n), r_cpu(n), t_cpu(n), r_gpu(n), b_ref(n))
    do k = 1, nz; do j = 1, ny; do i = 1, nx
        u(i + nx*(j - 1 + ny*(k - 1))) = sin(2*pi*gr(1)%nodes(i))*cos(4*pi*gr(2)%nodes(j))*sin(6*pi*gr(3)%nodes(k)) &
                                         + 0.1_wp*sin(real(37*i + 11*j + 5*k, wp))
        v(i + nx*(j - 1 + ny*(k - 1))) = cos(2*pi*gr(1)%nodes(i))*sin(2*pi*gr(3)%nodes(k)) + 0.1_wp*cos(real(3*i + 7*j + 13*k, wp))
    end do; end do; end do

    call TLab_AMD_Check(tlab_init(0_c_int), 'tlab_init')
    ! what the allocation hook does for q, s, txc, ...: device memory wrapped as Fortran arrays
    call TLab_AMD_Check(tlab_malloc(p_u, int(n, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_u, d_u, [n])
    call TLab_AMD_Check(tlab_malloc(p_v, int(n, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_v, d_v, [n])
    call TLab_AMD_Check(tlab_malloc(p_r, int(n, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_r, d_r, [n])
    call TLab_AMD_Check(tlab_malloc(p_t, int(n, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_t, d_t, [n])
    call TLab_AMD_Check(tlab_memcpy_h2d(p_u, c_loc(u), int(n, c_size_t)*8_c_size_t), 'h2d')
    call TLab_AMD_Check(tlab_memcpy_h2d(p_v, c_loc(v), int(n, c_size_t)*8_c_size_t), 'h2d')

    bcs = 0
    worst = 0.0_wp
    visc = 1.0_wp/5000.0_wp
    call OPR_Burgers_Initialize_AMD(visc, [1.0_wp])
    call OPR_Burgers_SetPlans(g)
    do ig = 1, 3
        ! OPR_Partial(OPR_P2_P1): same call, CPU module vs device drop-in
        select case (ig)
        case (1); call CPU_Partial_X(OPR_P2_P1, nx, ny, nz, bcs, g(1), u, r_cpu, t_cpu)
            call GPU_Partial_X(OPR_P2_P1, nx, ny, nz, bcs, g(1), d_u, d_r, d_t)
        case (2); call CPU_Partial_Y(OPR_P2_P1, nx, ny, nz, bcs, g(2), u, r_cpu, t_cpu)
            call GPU_Partial_Y(OPR_P2_P1, nx, ny, nz, bcs, g(2), d_u, d_r, d_t)
        case (3); call CPU_Partial_Z(OPR_P2_P1, nx, ny, nz, bcs, g(3), u, r_cpu, t_cpu)
            call GPU_Partial_Z(OPR_P2_P1, nx, ny, nz, bcs, g(3), d_u, d_r, d_t)
        end select
        call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_r, int(n, c_size_t)*8_c_size_t), 'd2h')
        err = maxval(abs(r_gpu - r_cpu))/maxval(abs(r_cpu)); worst = max(worst, err)
        print '(a,i1,a,es10.3)', 'OPR_Partial dir ', ig, ' (OPR_P2_P1) second derivative rel-err ', err
        call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_t, int(n, c_size_t)*8_c_size_t), 'd2h')
        err = maxval(abs(r_gpu - t_cpu))/maxval(abs(t_cpu)); worst = max(worst, err)
        print '(a,i1,a,es10.3)', 'OPR_Partial dir ', ig, ' (OPR_P2_P1) first derivative  rel-err ', err
        ! OPR_Burgers (U_IN) vs nu*P2 - v*P1 assembled from the CPU operators (vburgers.f90:78-151)
        b_ref = visc*r_cpu - v*t_cpu
        select case (ig)
        case (1); call OPR_Burgers_X(OPR_B_U_IN, 0, nx, ny, nz, bcs, d_u, d_v, d_r, d_t)
        case (2); call OPR_Burgers_Y(OPR_B_U_IN, 0, nx, ny, nz, bcs, d_u, d_v, d_r, d_t)
        case (3); call OPR_Burgers_Z(OPR_B_U_IN, 0, nx, ny, nz, bcs, d_u, d_v, d_r, d_t)
        end select
        call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_r, int(n, c_size_t)*8_c_size_t), 'd2h')
        err = maxval(abs(r_gpu - b_ref))/maxval(abs(b_ref)); worst = max(worst, err)
        print '(a,i1,a,es10.3)', 'OPR_Burgers dir ', ig, ' rel-err vs nu*P2 - u*P1 (CPU) ', err
    end do
    ! ---- OPR_Poisson through the drop-in module (procedure pointer of opr_elliptic.f90:33-46): forcing assembled with the reference's
    ! CPU operators as div(grad phi) with first derivatives (the construction of src/valid/elliptic/vpoisson.f90 / SURVEY 4.4), Neumann data
    ! = d(phi)/dy at the walls; the device solver must return that d(phi)/dy ----
    allocate (phi(n), dphidy(n), f(n), w1(n), w2(n), hb(nx*nz), ht(nx*nz))
    do k = 1, nz; do j = 1, ny; do i = 1, nx
        phi(i + nx*(j - 1 + ny*(k - 1))) = sin(2*pi*gr(1)%nodes(i))*cos(4*pi*gr(3)%nodes(k))*exp(0.5_wp*gr(2)%nodes(j)) &
                                           + cos(6*pi*gr(1)%nodes(i) + 1.0_wp)*gr(2)%nodes(j)**2 + 0.3_wp*sin(4*pi*gr(3)%nodes(k))*cos(2*gr(2)%nodes(j))
    end do; end do; end do
    call CPU_Partial_Y(OPR_P1, nx, ny, nz, bcs, g(2), phi, dphidy)
    call CPU_Partial_Y(OPR_P1, nx, ny, nz, bcs, g(2), dphidy, f)
    call CPU_Partial_X(OPR_P1, nx, ny, nz, bcs, g(1), phi, w1); call CPU_Partial_X(OPR_P1, nx, ny, nz, bcs, g(1), w1, w2); f = f + w2
    call CPU_Partial_Z(OPR_P1, nx, ny, nz, bcs, g(3), phi, w1); call CPU_Partial_Z(OPR_P1, nx, ny, nz, bcs, g(3), w1, w2); f = f + w2
    do k = 1, nz; do i = 1, nx
        hb(i + nx*(k - 1)) = dphidy(i + nx*(0 + ny*(k - 1)))
        ht(i + nx*(k - 1)) = dphidy(i + nx*(ny - 1 + ny*(k - 1)))
    end do; end do
    ntxc = (nx + 2)*ny*nz
    call TLab_AMD_Check(tlab_malloc(p_t1, int(ntxc, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_t1, d_t1, [ntxc])
    call TLab_AMD_Check(tlab_malloc(p_t2, int(ntxc, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_t2, d_t2, [ntxc])
    call TLab_AMD_Check(tlab_malloc(p_hb, int(nx*nz, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_hb, d_hb, [nx*nz])
    call TLab_AMD_Check(tlab_malloc(p_ht, int(nx*nz, c_size_t)*8_c_size_t), 'tlab_malloc'); call c_f_pointer(p_ht, d_ht, [nx*nz])
    call TLab_AMD_Check(tlab_memcpy_h2d(p_u, c_loc(f), int(n, c_size_t)*8_c_size_t), 'h2d')
    call TLab_AMD_Check(tlab_memcpy_h2d(p_hb, c_loc(hb), int(nx*nz, c_size_t)*8_c_size_t), 'h2d')
    call TLab_AMD_Check(tlab_memcpy_h2d(p_ht, c_loc(ht), int(nx*nz, c_size_t)*8_c_size_t), 'h2d')
    call OPR_Elliptic_Initialize_AMD(g, nx, ny, nz)
    call OPR_Poisson(nx, ny, nz, BCS_NN, d_u, d_t1, d_t2, d_hb, d_ht, d_r)
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_r, int(n, c_size_t)*8_c_size_t), 'd2h')
    err = maxval(abs(r_gpu - dphidy))/maxval(abs(dphidy)); worst = max(worst, err)
    print '(a,es10.3)', 'OPR_Poisson (drop-in pointer) dp/dy rel-err vs d(phi)/dy of the CPU operators ', err
    ! p itself is defined up to a constant: compare its x-derivative (CPU operator on the downloaded field) with that of phi
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_u, int(n, c_size_t)*8_c_size_t), 'd2h')
    call CPU_Partial_X(OPR_P1, nx, ny, nz, bcs, g(1), r_gpu, w1)
    call CPU_Partial_X(OPR_P1, nx, ny, nz, bcs, g(1), phi, w2)
    err = maxval(abs(w1 - w2))/maxval(abs(w2)); worst = max(worst, err)
    print '(a,es10.3)', 'OPR_Poisson (drop-in pointer) dp/dx rel-err ', err
    ! ---- EllipticOrder = CompactDirect6 (OPR_Poisson_FourierXZ_Direct): the host builds fdm_loc exactly as OPR_Elliptic_Initialize does
    ! (opr_elliptic.f90:107-124) and hands it over; Dirichlet data = phi at the walls, forcing = (d2/dx2 + d2/dy2 + d2/dz2) phi with the
    ! reference's CPU second-derivative operators (direct scheme in y): the discrete solution is phi itself ----
    fdm_loc%name = 'y'; fdm_loc%periodic = .false.; fdm_loc%uniform = .false.
    fdm_loc%der1%mode_fdm = FDM_COM6_DIRECT; fdm_loc%der2%mode_fdm = FDM_COM6_DIRECT
    call FDM_CreatePlan(gr(2), fdm_loc)
    call CPU_Partial_Y(OPR_P2, nx, ny, nz, bcs, fdm_loc, phi, f, w1)
    call CPU_Partial_X(OPR_P2, nx, ny, nz, bcs, g(1), phi, w2, w1); f = f + w2
    call CPU_Partial_Z(OPR_P2, nx, ny, nz, bcs, g(3), phi, w2, w1); f = f + w2
    do k = 1, nz; do i = 1, nx
        hb(i + nx*(k - 1)) = phi(i + nx*(0 + ny*(k - 1)))
        ht(i + nx*(k - 1)) = phi(i + nx*(ny - 1 + ny*(k - 1)))
    end do; end do
    call TLab_AMD_Check(tlab_memcpy_h2d(p_u, c_loc(f), int(n, c_size_t)*8_c_size_t), 'h2d')
    call TLab_AMD_Check(tlab_memcpy_h2d(p_hb, c_loc(hb), int(nx*nz, c_size_t)*8_c_size_t), 'h2d')
    call TLab_AMD_Check(tlab_memcpy_h2d(p_ht, c_loc(ht), int(nx*nz, c_size_t)*8_c_size_t), 'h2d')
    call OPR_Elliptic_Initialize_AMD(g, nx, ny, nz, fdm_loc)
    call OPR_Poisson(nx, ny, nz, BCS_DD, d_u, d_t1, d_t2, d_hb, d_ht, d_r)
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_u, int(n, c_size_t)*8_c_size_t), 'd2h')
    err = maxval(abs(r_gpu - phi))/maxval(abs(phi)); worst = max(worst, err)
    print '(a,es10.3)', 'OPR_Poisson (direct, BCS_DD) p rel-err vs phi ', err
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_r, int(n, c_size_t)*8_c_size_t), 'd2h')
    err = maxval(abs(r_gpu - dphidy))/maxval(abs(dphidy)); worst = max(worst, err)
    print '(a,es10.3)', 'OPR_Poisson (direct, BCS_DD) dp/dy rel-err vs d(phi)/dy of the CPU operators ', err
    ! ---- the SAME fdm_dt object built anew in place on another y grid (uniform nodes, where it was tanh-stretched): the drop-in must notice (its plan
    ! cache is keyed by the object's address AND a fingerprint of its tables) and serve the new tables, not the device plan of the old ones ----
    gr(2)%nodes = [(real(i - 1, wp)/real(ny - 1, wp), i=1, ny)]
    gr(2)%scale = gr(2)%nodes(ny) - gr(2)%nodes(1)
    if (allocated(g(2)%nodes)) deallocate (g(2)%nodes)
    if (allocated(g(2)%jac)) deallocate (g(2)%jac)
    if (allocated(g(2)%der1%lhs)) deallocate (g(2)%der1%lhs)
    if (allocated(g(2)%der1%rhs)) deallocate (g(2)%der1%rhs)
    if (allocated(g(2)%der1%lu)) deallocate (g(2)%der1%lu)
    if (allocated(g(2)%der1%mwn)) deallocate (g(2)%der1%mwn)
    if (allocated(g(2)%der2%lhs)) deallocate (g(2)%der2%lhs)
    if (allocated(g(2)%der2%rhs)) deallocate (g(2)%der2%rhs)
    if (allocated(g(2)%der2%lu)) deallocate (g(2)%der2%lu)
    if (allocated(g(2)%der2%mwn)) deallocate (g(2)%der2%mwn)
    g(2)%uniform = .true.
    call FDM_CreatePlan(gr(2), g(2))
    call TLab_AMD_Check(tlab_memcpy_h2d(p_u, c_loc(u), int(n, c_size_t)*8_c_size_t), 'h2d')
    call CPU_Partial_Y(OPR_P2_P1, nx, ny, nz, bcs, g(2), u, r_cpu, t_cpu)
    call GPU_Partial_Y(OPR_P2_P1, nx, ny, nz, bcs, g(2), d_u, d_r, d_t)
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(r_gpu), p_t, int(n, c_size_t)*8_c_size_t), 'd2h')
    err = maxval(abs(r_gpu - t_cpu))/maxval(abs(t_cpu)); worst = max(worst, err)
    print '(a,es10.3)', 'OPR_Partial_Y after FDM_CreatePlan on the same object, other grid: first derivative rel-err ', err
    print '(a,es10.3)', 'worst ', worst
    if (worst > 1.0e-11_wp) error stop 1
    print '(a)', 'dropin ok'
end program test_dropin

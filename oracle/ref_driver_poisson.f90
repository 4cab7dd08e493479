! TEST INFRASTRUCTURE ONLY -- placeholder, filled in with the Poisson (FDM_Int1 / OPR_ODE2) entry points.
subroutine ref_poisson_placeholder() bind(C, name='ref_poisson_placeholder')
end subroutine

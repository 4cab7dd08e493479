cd $GRAFT_REPO_ROOT
for NM in 4 8 4 8; do for G in 128; do echo "== TLAB_ODE_NM=$NM G=$G"; TLAB_ODE_NM=$NM TLAB_ODE_PAIR_XCD=$G TLAB_PROFILE_REPORT=1 python tools/bench_poisson.py --iters 7 2>&1 | grep -E "OPR_Poisson|k_ode_nn"; done; done

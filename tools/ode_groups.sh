cd $GRAFT_REPO_ROOT
for G in 0 16 32 64 128 256 512 1024; do echo "== TLAB_ODE_PAIR_XCD=$G"; TLAB_ODE_PAIR_XCD=$G TLAB_PROFILE_REPORT=1 python tools/bench_poisson.py --iters 7 2>&1 | grep -E "OPR_Poisson|k_ode_nn "; done

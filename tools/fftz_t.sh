cd $GRAFT_REPO_ROOT
for T in 8 16 8 16; do echo "== TLAB_FFTZ_T=$T"; TLAB_FFTZ_T=$T TLAB_PROFILE_REPORT=1 python tools/bench_poisson.py --iters 7 2>&1 | grep -E "OPR_Poisson|k_fftz|k_ode_nn"; done

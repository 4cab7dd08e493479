cd $GRAFT_REPO_ROOT
for G in 0 16 128; do
  export TLAB_ODE_PAIR_XCD=$G
  bash tools/pmc_any.sh gpurun_out/r06_pmc_ode_G$G.txt TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B > /dev/null 2>&1
  echo "== G=$G"; grep k_ode_nn gpurun_out/r06_pmc_ode_G$G.txt
done

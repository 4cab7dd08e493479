cd $GRAFT_REPO_ROOT
D=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0, "tools")
import fortran_host as F
F.write_case("$D", 512, 512, 512, 5, 1e-3)
PY
cd $D
for i in 1 2 3 4; do for P in 4 8; do
  rm -f tlab.log tlab.err
  TLAB_AMD_PLACE=$P TLAB_AMD_TIMING=1 $GRAFT_REPO_ROOT/tlab_amd/fortran/_build_rk/test_rk_driver > /dev/null 2>&1
  echo "TLAB_AMD_PLACE=$P: $(grep -o 'ms_per_substep.*' tlab.log)  $(grep -o 'PLACEMENT.*' tlab.log)"
done; done
rm -rf $D

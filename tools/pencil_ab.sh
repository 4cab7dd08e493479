cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pencil.py tests/test_gpu_fortran_dropin.py -m gpu -x -q 2>&1 | tail -4
for F in 1 0 1 0; do TLAB_PENCIL_FINAL=$F python bench.py --decomp 2x4 --slab-driver native --steps 6 --warmup 2 --cpu-sample 0 --no-freeslip-leg --no-fortran-host 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('TLAB_PENCIL_FINAL=$F ms_per_step %.3f' % r['ms_per_step'])
"; done

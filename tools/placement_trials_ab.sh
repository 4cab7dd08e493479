cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for T in "16 40" "32 56" "64 72"; do set -- $T; python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-freeslip-leg --no-fortran-host --placement-trials $1 --placement-pool $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); p = r['placement']; print('trials $1 pool $2: ms_per_step %.3f  search first %.2f best %.2f median %.2f  %.1f s' % (r['ms_per_step'], p['ms_first'], p['ms_best'], p['ms_median'], p['seconds']))
"; done; done

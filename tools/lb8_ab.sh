cd $GRAFT_REPO_ROOT
run() { python bench.py --loopback 8 --steps 12 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-28s ms_per_step %.3f' % ('$1', r['ms_per_step']), ' '.join('%s=%.2f' % (k['kernel'], k['total_ms']/k['calls']*k['calls']/6 if False else k['avg_ms']*k['calls']/max(1,r.get('steps',1))) for k in r['kernels'][:0]))
        ks = {k['kernel']: k for k in r['kernels']}
        tot = sum(k['avg_ms']*k['calls'] for k in r['kernels'])
        n = None
        for k in ('k_htile<BURGERS>', 'k_zslab<BURGERS,B>', 'k_zslab<BURGERS,A>', 'k_copy_blocks', 'k_ode_nn', 'k_xline<BURGERS>'):
            if k in ks: print('    %-22s calls %4d  avg %.1f us' % (k, ks[k]['calls'], ks[k]['avg_ms']*1e3))
"; }
run default
TLAB_HTILE_LINES=16 run TLAB_HTILE_LINES=16
run default
TLAB_HTILE_LINES=16 run TLAB_HTILE_LINES=16

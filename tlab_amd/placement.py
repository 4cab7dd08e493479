"""Where the arrays of a driver live.  The rate of a kernel that streams many arrays at once depends on WHICH device allocations they
are -- not on any one of them, on the set (tools/placement_probe).  The single-domain driver searches (tlab_dns_place_arrays, Dns.place_arrays); the
decomposed drivers, whose ranks hold 17 smaller arrays each, draw them at random from a pool: there the allocator's own order (arrays allocated one after
the other) costs 1.3 % of the substep and the spread among random draws is 0.7 % (tools/placement_slab_probe.py), so one draw does what a search would."""
import numpy as np

from .lib import load, check, c_vp


def redraw_rank_arrays(d, bind, pool=34, seed=0):
    """d: NativeSlabDns / NativePencilDns (st[rank][name] lists of tensors, local_ranks, nscal, n, isize_txc, _h); bind: name of the C entry point that
    takes (handle, local index, q, s, hq, hs, txc).  Every local rank's arrays move to allocations drawn at random from `pool` fresh ones of the txc
    size; the values come along; the allocations not drawn and the old arrays go back to the allocator."""
    import torch
    rng = np.random.default_rng(seed)
    fn = getattr(load(), bind)
    nroles = 2 * (3 + d.nscal) + 9
    pool = max(int(pool), nroles)
    for l, r in enumerate(d.local_ranks):
        S = d.st[r]
        dev = S["q"][0].device
        cand = [torch.zeros(d.isize_txc, dtype=torch.float64, device=dev) for _ in range(pool)]
        pick = [cand[i] for i in rng.permutation(pool)[:nroles]]
        new, pos = {}, 0
        for name, cnt, m in (("q", 3, d.n), ("s", d.nscal, d.n), ("hq", 3, d.n), ("hs", d.nscal, d.n), ("txc", 9, d.isize_txc)):
            new[name] = [t[:m] for t in pick[pos:pos + cnt]]
            pos += cnt
            for a, b in zip(new[name], S[name]):
                a.copy_(b)
        arr = lambda ts: (c_vp * max(len(ts), 1))(*[t.data_ptr() for t in ts])       # noqa: E731
        check(fn(d._h, l, arr(new["q"]), arr(new["s"]), arr(new["hq"]), arr(new["hs"]), arr(new["txc"])), bind)
        d.st[r] = new
        del cand, pick, S
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return {"kind": "every rank's arrays drawn at random from a pool of %d fresh allocations (no search)" % pool, "pool": pool, "seed": int(seed)}

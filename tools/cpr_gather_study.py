#!/usr/bin/env python3
"""CPR-BiCGStab iteration counts of a decomposed run on the CPU (oracle only): one domain, one hierarchy per subdomain with nothing
between them (opmhip_config.cpr_gather_rows < 0), and the pressure stage that spans the subdomains (orc_cpr_solve_blocks with
gather_rows: level 0 smooths with the whole system's operator, its residual goes down to each subdomain's first level of at most
`rows` rows, those levels are joined and cycled on as one system) for several values of `rows`.
    python tools/cpr_gather_study.py N WORLD TOL ROWS...        e.g.  20 8 1e-4 40 600 2500
What the design of csrc/cpr.hip: cpr_gathered_cycle rests on (numbers of round 4, heterogeneous synthetic case, 20-day step):
    8 x 20^3, 1e-4:  one domain 12.0   per subdomain 56.5   rows 40: 19.0 (508 joined rows)   600: 13.0 (2 420)   2 500: 11.5 (13 604)
    8 x 30^3, 1e-4:  one domain 13.5   per subdomain 48.0   rows 120: 20.5 (1 398)   1 000: 14.5 (5 028)   4 000: 15.0 (22 114)
- joined levels whose aggregates hold up to ~50 cells cost nothing against one domain; the smoothing of the levels between level 0 and
the joined one is not missed (level 0's and the block ILU0's are what count).  Variants measured and dropped: the subdomains' own
cycles above the joined level with the couplings between subdomains left out of them (inconsistent residuals at the interfaces: worse
than no joined level at all), a coarse correction added to or multiplied with complete per-subdomain cycles (both correct the smooth
part twice), a post-smoothing residual without the neighbours' pressure (69 instead of 20 iterations with an otherwise exact stage)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
pkg = importlib.import_module("opm-autodiff_amd")
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
n, world, tol = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rows = [int(v) for v in sys.argv[4:]]
px, py, pz = pkg.ras.block_layout(world)
g = pkg.decks.cartesian_case(px * n, py * n, pz * n, state="mixed", heterogeneous=True)
owner = np.asarray(pkg.ras.cartesian_owner(px * n, py * n, pz * n, px, py, pz), np.int32)
src = pkg.decks.five_spot_source(g, rate_sm3_per_day=60.0)
o = oracle_bind.OracleModel(orc, g); o.set_state(g["pv"], g["meaning"]); o.set_source(src)
jac, res = o.assemble(20 * 86400.0, 0)
Nb, rp, ci = g["Nb"], g["rowptr"], g["col"]
_, one = oracle_bind.OracleCpr(orc).solve(Nb, rp, ci, jac, res, tol=tol)
_, alone, _ = orc.cpr_solve_blocks(Nb, rp, ci, jac, res, owner, tol=tol)
print("%d x %d^3, tol %g: one domain %.1f, one hierarchy per subdomain %.1f" % (world, n, tol, one.it, alone.it), end="")
for r in rows:
    _, rr, lev, glev, _ = orc.cpr_solve_blocks(Nb, rp, ci, jac, res, owner, tol=tol, gather_rows=r)
    print(" | rows %d: %.1f (joined levels %s)" % (r, rr.it, glev), end="")
print()

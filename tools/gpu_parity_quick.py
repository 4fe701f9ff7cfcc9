import sys, time
sys.path.insert(0, '.')
import numpy as np
import voronoirt_amd as v
from voronoirt_amd import synth
from oracle import oracle as orc

for name, (pos,nbr,b) in {"bcc": synth.bcc_grid(8,12,seed=2), "voro": synth.voronoi_grid(3000, seed=5, bounds=(0,2.0,0,1,0,1), scale_height=0.7)}.items():
    so = orc.make_sites(pos,nbr,b)
    hs = v.VoronoiSites(pos,nbr,b,device=0)
    lines = hs.Delaunay_lines
    ol = np.nan_to_num(so.delaunay_lines, nan=0.0)
    # wall slots: oracle leaves nan->0 ; product 0
    mask = np.zeros(ol.shape[:2], bool)
    for j in range(so.D):
        mask[:, j] = (so.neighbours[j+1] > 0) & (j < so.neighbours[0])
    print(name, "lines bit-exact:", np.array_equal(lines[mask], so.delaunay_lines[mask]))
    n = so.n
    rng = np.random.default_rng(0)
    S = 1 + rng.random(n)
    alpha = 10**rng.uniform(-3, 3, n) / (b[3]-b[2]) * 10
    wq, th, ph, nq = v.read_quadrature('ul7n12.dat')
    ks = v.quadrature_directions(th, ph)
    plan = v.FormalPlan(hs, ks, 3)
    print(name, "plan levels", plan.num_levels, "nodes", plan.num_nodes)
    worst = 0
    for a,(t,p) in enumerate(zip(th,ph)):
        k = orc.direction(t,p)
        assert np.array_equal(k, ks[a])
        up, dots, w, r, st = orc.upwind_table(so, k)
        gup, gd, gw, gr = plan.upwind(a)
        ok = st == 0
        assert np.array_equal(up[ok], gup[ok]), ("upwind ids differ", a)
        assert np.array_equal(dots[ok], gd[ok]), "dots differ"
        assert np.array_equal(r[ok], gr[ok]), "r differ"
        werr = np.abs(gw[ok]-w[ok]).max()
        dirn = 1 if t>90 else -1
        lay = so.layers_up if dirn>0 else so.layers_down
        I0 = rng.random(lay[1]-1)
        ref = (orc.Delaunay_upII if dirn>0 else orc.Delaunay_downII)(k,S,I0,alpha,so,3)
        got = (v.Delaunay_upII if dirn>0 else v.Delaunay_downII)(k,S,I0,alpha,hs,3)
        err = np.abs(got-ref).max()/np.abs(ref).max()
        worst = max(worst, err)
        print(name, "theta %.1f werr %.1e Ierr %.2e" % (t, werr, err))
    # J with nlam
    nlam = 5
    S2 = 1 + rng.random((n,nlam)); al2 = alpha[:,None]*(1+rng.random((n,nlam)))
    I0u = rng.random((so.layers_up[1]-1, nlam))
    Jref = orc.J_voronoi(wq, th, ph, S2, al2, so, I0_up=I0u, nthreads=4)
    Jgot = v.J_lambda_voronoi(S2, al2, hs, 'ul7n12.dat', I0_up=I0u)
    print(name, "J err", np.abs(Jgot-Jref).max()/np.abs(Jref).max())
    al3 = np.stack([al2*(1+0.1*a) for a in range(nq)])
    Jref = orc.J_voronoi(wq, th, ph, S2, al3, so, I0_up=I0u, nthreads=4)
    Jgot = v.J_lambda_voronoi(S2, al3, hs, 'ul7n12.dat', I0_up=I0u)
    print(name, "J (per-angle alpha) err", np.abs(Jgot-Jref).max()/np.abs(Jref).max())
    assert worst < 1e-10
print("OK")

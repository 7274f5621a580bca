"""NumPy feasibility probe for the windowed fine-grid absorption (k_absorb_win): relative error of interpolating the
window-far O2 line sums from n Chebyshev nodes, by window size (G chunks of 16), margin and node count, on the oracle's
own line formulas over the 20-60 GHz grid (all 180 levels of a synthetic profile, incl. the sharp-line top levels).
    python tools/window_probe.py        # CPU only
"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter("ignore")
from mwr_fast_forward_operators_and_lbls_amd import spectroscopy as sp, profiles as pr
from oracle import lbl_oracle as lo
m = sp.get_model("R24")
P = pr.synthetic_profiles(1, 5)
p, t, rh = P["p"][0], P["t"][0], P["rh"][0]
e, rho = lo.vapor(t, rh)
frq = pr.fine_grid_frequencies(1000)

def o2_terms(m, p, t, e, f, lines):
    """sum over selected lines of the O2 line terms (without the final scale), per level, for frequency array f -> [nlev, nf]"""
    th = 300.0/t; th1 = th-1; b = th**m.o2_x
    rvap = (0.01*8.314510)/18.01528
    ekpa = e/10; pdry = p/10-ekpa; pres=(pdry+ekpa)*10; vapden = ekpa*10/(rvap*t)
    preswv = vapden*t/m.o2_pvap_div; presda = pres-preswv
    den = 0.001*(presda*b + m.o2_wv_factor*preswv*th); pe2 = den*den
    L = m.o2
    out = np.zeros((len(t), len(f)))
    for k in lines:
        y = den*(L["y0"][k]+L["y1"][k]*th1); dnu = pe2*(L["dnu0"][k]+L["dnu1"][k]*th1)
        g = 1+pe2*(L["g0"][k]+L["g1"][k]*th1); df = L["w300"][k]*den
        s = L["s300"][k]*np.exp(-L["be"][k]*th1)
        d1 = f[None,:]-L["f"][k]-dnu[:,None]; d2 = f[None,:]+L["f"][k]+dnu[:,None]
        sf1 = (df[:,None]*g[:,None]+d1*y[:,None])/(d1*d1+df[:,None]**2)
        sf2 = (df[:,None]*g[:,None]-d2*y[:,None])/(d2*d2+df[:,None]**2)
        out += s[:,None]*(sf1+sf2)*(f[None,:]/L["f"][k])**2
    return out

def cheb_nodes(lo_, hi_, n):
    k = np.arange(n); x = np.cos(np.pi*(2*k+1)/(2*n))
    return 0.5*(lo_+hi_) + 0.5*(hi_-lo_)*x

def lagrange_matrix(nodes, targets):
    # barycentric weights
    n = len(nodes); w = np.ones(n)
    for j in range(n):
        w[j] = 1.0/np.prod(nodes[j]-np.delete(nodes,j))
    M = np.zeros((len(targets), n))
    for i, x in enumerate(targets):
        d = x - nodes
        if np.any(d == 0):
            M[i, np.argmin(np.abs(d))] = 1; continue
        q = w/d; M[i] = q/q.sum()
    return M

fc = m.o2["f"]
for G in (8, 16):
  for margin in (2.0, 3.0, 4.0):
    for n in (16, 20, 24, 28):
        worst = 0; nnear = []
        for w0 in range(0, 1000, 16*G):
            tg = frq[w0:w0+16*G]
            flo, fhi = tg[0], tg[-1]
            near = [k for k in range(len(fc)) if flo - margin <= fc[k] <= fhi + margin]
            far = [k for k in range(len(fc)) if k not in near]
            nnear.append(len(near))
            nodes = cheb_nodes(flo, fhi, n)
            S_nodes = o2_terms(m, p, t, e, nodes, far)
            S_true = o2_terms(m, p, t, e, tg, far)
            S_all = o2_terms(m, p, t, e, tg, range(len(fc)))
            M = lagrange_matrix(nodes, tg)
            S_int = S_nodes @ M.T
            # error relative to the TOTAL O2 line sum (what matters for absorption)
            err = np.abs(S_int - S_true)/np.abs(S_all)
            worst = max(worst, err.max())
        print(f"G={G:2d} ({16*G} freqs) margin={margin} n={n}: worst rel err vs total O2 sum {worst:.2e}  near lines avg {np.mean(nnear):.1f} max {max(nnear)}")

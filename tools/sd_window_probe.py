"""Would the speed-dependent 22-GHz line interpolate across a 128-frequency window?  (CPU, oracle only.)

For configs[4] grids: per window, the levels whose whole window sits inside the line's SD region, whether a
whole wave (64 levels) does, and the Chebyshev-16 interpolation error of the line's term there.  Result
(profiles/r02_sd_window_probe.txt): accurate to 1e-12 once the window is >=8 GHz from the centre, but only
3 of 24 (wave, window) pairs qualify, so the windowed kernel keeps evaluating the SD line directly."""
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
L = m.h2o
def sd_line_term(f, i=0):
    """full contribution s*res of SD line i (both Lorentz terms + SD shape where inner) per level x freq, and inner mask"""
    rvap = (0.01*8.314510)/18.01528
    ekpa = e/10; pdry = p/10-ekpa; pp=(pdry+ekpa)*10; rho_=ekpa*10/(rvap*t)
    pvap = rho_*t/m.h2o_pvap_div; pda = pp-pvap
    ti = m.h2o_reftline/t; tiln=np.log(ti); ti2=np.exp(2.5*tiln)
    w0 = L["w0"][i]*pda*ti**L["x"][i] + L["w0s"][i]*pvap*ti**L["xs"][i]
    w2 = L["w2"][i]*pda*ti**L["xw2"][i] + L["w2s"][i]*pvap*ti**L["xw2s"][i]
    d2_ = L["d2"][i]*pda + L["d2s"][i]*pvap
    shift = L["sh"][i]*pda*(1-L["aair"][i]*tiln)*ti**L["xh"][i] + L["shs"][i]*pvap*(1-L["aself"][i]*tiln)*ti**L["xhs"][i]
    s = L["s1"][i]*ti2*np.exp(L["b2"][i]*(1-ti))
    wsq=w0**2; base=w0/(562500+wsq)
    out = np.zeros((len(t), len(f))); inner_all = np.zeros((len(t), len(f)), bool)
    for j, ff in enumerate(f):
        df0 = ff - L["fl"][i] - shift; df1 = ff + L["fl"][i] + shift
        lor0 = np.where(np.abs(df0)<750, w0/(df0**2+wsq)-base, 0.0)
        lor1 = np.where(np.abs(df1)<750, w0/(df1**2+wsq)-base, 0.0)
        use = (w2>0)&(np.abs(df0)<10*w0)
        xc = ((w0-1.5*w2)+1j*(df0+1.5*d2_))/(w2-1j*d2_)
        xrt = np.sqrt(xc)
        pxw = 1.77245385090551603*xrt*lo.dcerror(-np.imag(xrt), np.real(xrt))
        sdv = 2*(1-pxw)/(w2-1j*d2_)
        lor0 = np.where(use, np.real(sdv)-base, lor0)
        out[:, j] = s*(lor0+lor1)*(ff/L["fl"][i])**2
        inner_all[:, j] = use
    return out, inner_all
def cheb_nodes(lo_, hi_, n):
    k = np.arange(n); x = np.cos(np.pi*(2*k+1)/(2*n)); return 0.5*(lo_+hi_)+0.5*(hi_-lo_)*x
def lag(nodes, targets):
    n=len(nodes); w=np.array([1.0/np.prod(nodes[j]-np.delete(nodes,j)) for j in range(n)])
    M=np.zeros((len(targets),n))
    for i,x in enumerate(targets):
        q=w/(x-nodes); M[i]=q/q.sum()
    return M
# total wet line sum for scale: use all lines' terms ~ approximate by SD line + others via oracle absorption awet
aw = np.array([lo.clearsky_absorption(m, p, t, e, ff)[0] for ff in frq]).T   # [lev, f] Np/km
# convert line term to absorption units: awet = 3.183e-5*den*sum*... too fiddly: compare relative to the SD line term itself and report its share
for w0i in range(0, 1000, 128):
    tg = frq[w0i:w0i+128]
    nodes = cheb_nodes(tg[0], tg[-1], 16)
    tn, inn = sd_line_term(nodes); tt, int_ = sd_line_term(tg)
    M = lag(nodes, tg)
    ti_ = tn @ M.T
    # levels where the whole window (nodes and targets) is inner, per wave of 64 levels all must be inner
    ok_lev = inn.all(axis=1) & int_.all(axis=1)
    dist = min(abs(tg[0]-22.235), abs(tg[-1]-22.235)) if not (tg[0] <= 22.235 <= tg[-1]) else 0.0
    if ok_lev.any():
        err = np.abs(ti_[ok_lev]-tt[ok_lev])/np.abs(tt[ok_lev])
        print(f"window {tg[0]:.2f}-{tg[-1]:.2f} GHz  dist to centre {dist:.1f}: levels fully inner {ok_lev.sum():3d} (first {np.argmax(ok_lev)}, waves fully inner: {[int(ok_lev[64*w:min(64*w+64,180)].all()) for w in range(3)]})  max rel err of the SD-line term {err.max():.2e}")
    else:
        print(f"window {tg[0]:.2f}-{tg[-1]:.2f}: no fully-inner level")

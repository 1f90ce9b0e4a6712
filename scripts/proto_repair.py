"""Prototype (CPU, numpy): active-set repair of the storage solve — all segments at once, contact set updated
between rounds — measured against the oracle's exact mode on a scaled config2 trajectory.
Not product code; used to choose the update rule before writing the HIP version."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import __graft_entry__ as ge

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 40
RMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ADD_ALL = int(sys.argv[4]) if len(sys.argv) > 4 else 0

GMUL = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
RAND = int(sys.argv[6]) if len(sys.argv) > 6 else 0
CHAIN_RESET = int(sys.argv[7]) if len(sys.argv) > 7 else 1
COLD_E = int(sys.argv[8]) if len(sys.argv) > 8 else 0
FWD = int(os.environ.get("FWD", "0"))
FWD_MAXREL = int(os.environ.get("FWD_MAXREL", "2"))
pp = synth.baseline_config(2, scale=scale)
if RAND:
    rng = np.random.default_rng(5)
    pp.sto_mc = rng.uniform(0.5, 3.5, pp.S); pp.sto_pmax = rng.uniform(5, 20, pp.S); pp.sto_emax = pp.sto_pmax * rng.uniform(0.7, 4.0, pp.S)
A = pp.G + pp.S
gam, w = GMUL / A, 1.0
from oracle.binding import OracleApi, set_threads
api = OracleApi(ge.ORACLE_LIB)
eng = _capi.Engine(api, params=_capi.default_params(gamma=gam, eps=0.0), mode=1, **pp.engine_kwargs())
set_threads(eng, 8)
T, S = pp.T, pp.S
a0 = w + gam
idet0 = 1.0 / (a0 * a0 - gam * gam)
s20 = 2.0 / (a0 + gam)


def box2(rD, rC, pm):
    Df = np.clip((a0 * rD + gam * rC) * idet0, 0.0, pm)
    Cf = (rC + gam * Df) / a0
    Dlo = np.clip(rD / a0, 0.0, pm)
    Dhi = np.clip((rD + gam * pm) / a0, 0.0, pm)
    C = np.clip(Cf, 0.0, pm)
    D = np.where(Cf < 0.0, Dlo, np.where(Cf > pm, Dhi, Df))
    fD = (D > 0) & (D < pm)
    fC = (C > 0) & (C < pm)
    sg = np.where(fD & fC, s20, np.where(fD | fC, 1.0 / a0, 0.0))
    return D, C, sg


class Sto:
    def __init__(self, mc, pm, em):
        self.mc, self.pm, self.em = mc, pm, em
        self.nu = None          # per-step prices of the last accepted solve
        self.kind = None


def segments(kind):
    """list of (a, e, kind_e) with e inclusive; last segment may be open (kind 0)"""
    segs, a = [], 0
    for t in range(T):
        if kind[t] != 0 or t == T - 1:
            segs.append((a, t, kind[t]))
            a = t + 1
    return segs


def solve_round(st, rD0, rC0, kind, nu_step, stats):
    """segment solve for a fixed contact set. returns dict with D, C, E, seg prices, ok flags"""
    pm, em = st.pm, st.em
    segs = segments(kind)
    D = np.zeros(T); C = np.zeros(T); nus = np.zeros(T); sgs = np.zeros(T)
    seginfo = []
    base = 0.0
    newton_ok = True
    depth = 0
    for (a, e, kd) in segs:
        sl = slice(a, e + 1)
        if kd == 0:
            nu = 0.0
            d, c, sg = box2(rD0[sl] - nu, rC0[sl] + nu, pm)
            flat = False
        else:
            tgt = 0.0 if kd == 1 else em
            nu = nu_step[e]
            lo, hi = -np.inf, np.inf
            flat = False
            for itn in range(60):
                d, c, sg = box2(rD0[sl] - nu, rC0[sl] + nu, pm)
                r = base + (c - d).sum() - tgt
                if abs(r) <= 1e-12 * (1 + em):
                    break
                if r < 0: lo = nu
                else: hi = nu
                ssum = sg.sum()
                stats['newton'] += 1
                if ssum > 0:
                    trial = nu - r / ssum
                else:
                    # flat: jump just past nearest kink in the right direction
                    dirn = 1.0 if r < 0 else -1.0
                    psc_shift = gam * (d - c)   # theta + gam(dd-cc) - theta
                    # kinks: prices at which D or C leaves its bound, net injection frozen
                    bD = rD0[sl] + gam * c      # D leaves 0 when nu < rD0 + gam C ... use closed forms
                    cand = np.concatenate([rD0[sl] + gam * c, rD0[sl] - a0 * pm + gam * c,
                                           -rC0[sl] - gam * d, a0 * pm - gam * d - rC0[sl]])
                    dist = (cand - nu) * dirn
                    dist = dist[dist > 1e-15 * (1 + abs(nu))]
                    if dist.size == 0:
                        newton_ok = False
                        break
                    best = dist.min()
                    trial = nu + dirn * (best + 1e-9 * (1 + abs(nu) + best))
                    stats['flatjump'] += 1
                if not (trial > lo and trial < hi):
                    if np.isfinite(lo) and np.isfinite(hi):
                        trial = 0.5 * (lo + hi)
                        if not (trial > lo and trial < hi):
                            break
                nu = trial
            else:
                newton_ok = False
            depth = max(depth, itn + 1)
            flat = sg.sum() == 0.0
        D[sl], C[sl], nus[sl], sgs[sl] = d, c, nu, sg
        # corner interval of the segment (valid prices when flat)
        lo_i = np.full(e - a + 1, -np.inf); hi_i = np.full(e - a + 1, np.inf)
        lo_i = np.where(d <= 0, rD0[sl] + gam * c, lo_i)
        hi_i = np.where(d >= pm, rD0[sl] - a0 * pm + gam * c, hi_i)
        hi_i = np.where(c <= 0, np.minimum(hi_i, -rC0[sl] - gam * d), hi_i)
        lo_i = np.where(c >= pm, np.maximum(lo_i, a0 * pm - gam * d - rC0[sl]), lo_i)
        seginfo.append(dict(a=a, e=e, kind=kd, nu=nu, flat=flat, lo=lo_i.max(), hi=hi_i.min(), base=base))
        base = base + (c - d).sum() if kd == 0 else (0.0 if kd == 1 else em)
    E = np.cumsum(C - D)
    stats['depth'] += depth
    return D, C, E, nus, seginfo, newton_ok


def certificate(st, seginfo, E):
    """returns (ok, list of contact steps to release, list of (step, kind) to add)"""
    em = st.em
    tolE = 1e-11 * (1 + em)
    n = len(seginfo)
    # feasible intervals right to left
    flo, fhi = 0.0, 0.0
    release = []
    for i in range(n - 1, -1, -1):
        sg = seginfo[i]
        if sg['kind'] == 0:
            flo, fhi = 0.0, 0.0
            continue
        if sg['flat'] and sg['lo'] <= sg['hi'] and sg['nu'] >= sg['lo'] - 1e-9 and sg['nu'] <= sg['hi'] + 1e-9:
            mlo, mhi = sg['lo'], sg['hi']
        else:
            mlo = mhi = sg['nu']
        nlo = max(mlo, flo) if sg['kind'] == 1 else mlo
        nhi = min(mhi, fhi) if sg['kind'] == 2 else mhi
        tn = 1e-10 * (1 + min(abs(nlo), abs(nhi)))
        if nlo > nhi + tn:
            release.append(sg['e'])
            if CHAIN_RESET: flo, fhi = mlo, mhi
            else: flo, fhi = nlo, nhi
        else:
            flo, fhi = nlo, max(nlo, nhi)
        sg['chosen'] = min(max(sg['nu'], flo), fhi)
    if FWD and (FWD < 4 or 0 < len(release) <= FWD_MAXREL):
        # forward pass (left to right): the interval a segment's price may take given everything to its LEFT; an empty
        # interval releases the contact in front of the segment, and the inverted bound travels on (a run of idle contacts
        # that a cheaper segment on its left wants to charge through is released in ONE round instead of one contact per round)
        plo, phi = -np.inf, np.inf
        prev_end = None
        for i, sg in enumerate(seginfo):
            if sg['kind'] != 0 and sg['flat'] and sg['lo'] <= sg['hi'] and sg['nu'] >= sg['lo'] - 1e-9 and sg['nu'] <= sg['hi'] + 1e-9:
                mlo, mhi = sg['lo'], sg['hi']
            elif sg['kind'] == 0:
                mlo = mhi = 0.0
            else:
                mlo = mhi = sg['nu']
            glo, ghi = max(mlo, plo), min(mhi, phi)
            tn = 1e-10 * (1 + min(abs(glo), abs(ghi)))
            if glo > ghi + tn and prev_end is not None and (FWD in (2, 4) or sg['a'] == sg['e']):
                if prev_end not in release: release.append(prev_end)
            if sg['kind'] == 1: plo, phi = -np.inf, ghi
            elif sg['kind'] == 2: plo, phi = glo, np.inf
            prev_end = sg['e']
    add = []
    for sg in seginfo:
        a, e = sg['a'], sg['e']
        seg = E[a:e + 1]
        under = -seg
        over = seg - em
        if ADD_ALL:
            for t in range(a, e + 1):
                if E[t] < -tolE: add.append((t, 1))
                elif E[t] > em + tolE: add.append((t, 2))
        else:
            viol = np.maximum(under, over)
            vm = viol.max()
            if vm > tolE:
                for t in range(a, e + 1):
                    if viol[t - a] == vm: add.append((t, 1 if under[t - a] >= over[t - a] else 2))
    return (not release and not add), release, add


def kinds_from_E(E, em):
    tolc = 1e-9 * (1 + em)
    return np.where(E <= tolc, 1, np.where(E >= em - tolc, 2, 0)).astype(int)


def solve_storage(st, D0, C0, th0, stats):
    q0 = D0 - C0
    theta = th0 - gam * q0
    rD0 = w * D0 - st.mc - theta
    rC0 = w * C0 - st.mc + theta
    if st.nu is None and COLD_E:
        kind = kinds_from_E(np.cumsum(C0 - D0), st.em)
        nu_step = np.zeros(T)
    elif st.nu is None:
        # cold: structure from a clamped forward pass at nu = 0
        d, c, _ = box2(rD0, rC0, st.pm)
        e = 0.0
        kind = np.zeros(T, int)
        for t in range(T):
            e2 = e + c[t] - d[t]
            if e2 <= 0: kind[t] = 1; e2 = 0.0
            elif e2 >= st.em: kind[t] = 2; e2 = st.em
            e = e2
        nu_step = np.zeros(T)
    else:
        kind = kinds_from_E(np.cumsum(C0 - D0), st.em)
        nu_step = st.nu.copy()
    trace = []
    for rnd in range(RMAX):
        D, C, E, nus, seginfo, nok = solve_round(st, rD0, rC0, kind, nu_step, stats)
        ok, release, add = certificate(st, seginfo, E)
        trace.append((np.flatnonzero(kind).tolist(), [int(kind[t]) for t in np.flatnonzero(kind)], release, add,
                      [round(sg['nu'], 3) for sg in seginfo]))
        st.trace = trace
        if ok and nok:
            for sg in seginfo:
                if sg['kind'] != 0:
                    nus[sg['a']:sg['e'] + 1] = sg['chosen']
            st.nu = nus
            return D, C, rnd + 1
        nu_step = nus
        for t in release: kind[t] = 0
        for (t, kd) in add: kind[t] = kd
        if not release and not add:
            break
    st.nu = None
    return None, None, RMAX + 1


stos = [Sto(pp.sto_mc[i], pp.sto_pmax[i], pp.sto_emax[i]) for i in range(S)]
for k in range(1, NIT + 1):
    _, D0, C0, _ = eng.get_primal()
    lam = eng.get_duals()[0]
    inj = eng.get_consensus()[0]
    s = inj.sum(axis=0)
    th0 = lam + gam * s
    eng.iterate(1)
    _, Dn, Cn, En = eng.get_primal()
    hist = np.zeros(RMAX + 2, int)
    stats = dict(newton=0, flatjump=0, depth=0)
    worst = 0.0
    depths = []
    for i, st in enumerate(stos):
        stats['depth'] = 0
        D, C, r = solve_storage(st, D0[i], C0[i], th0, stats)
        hist[r] += 1
        depths.append(stats['depth'])
        if r >= 5 and k >= 4 and os.environ.get("TRACE"):
            print(f"  storage {i} (pm {st.pm}, em {st.em}) took {r} rounds:")
            for j, (pos, kd, rel, add, nus_) in enumerate(st.trace):
                print(f"    round {j}: contacts {list(zip(pos, kd))} prices {nus_} -> release {rel} add {add}")
        if D is not None:
            worst = max(worst, np.abs(D - Dn[i]).max(), np.abs(C - Cn[i]).max())
    print(f"it {k}: rounds hist {hist[1:].tolist()} (last = failed) newton/sto {stats['newton']/S:.1f} flatjumps/sto {stats['flatjump']/S:.2f} worst diff {worst:.2e} | eval depth mean {np.mean(depths):.1f} max {np.max(depths)}", flush=True)

"""The lane-cooperative formulations of RVO2's programs (crowdnav_amd/csrc/orca_device.h: lp_planar_coop,
lp_relaxed_coop) against the sequential programs, both emulated in numpy float32 one operation at a time: a lane per
(agent, half-plane), rounds of "first violated line -> every lane's bound on it -> fold in line order".  The device code
is checked against the oracle on the GPU; this pins the ALGORITHM (order of operations, tie handling, early exits) on
random programs incl. parallel lines and infeasible ones, bit for bit, without a GPU."""
import numpy as np

f = np.float32
EPS = f(1e-5)
INF = f(np.inf)

def det(ax,ay,bx,by): return f(f(ax*by)-f(ay*bx))
def start(radius,ox,oy):
    if f(f(ox*ox)+f(oy*oy)) > f(radius*radius):
        inv=f(f(1.0)/np.sqrt(f(f(ox*ox)+f(oy*oy))))
        return f(f(ox*inv)*radius), f(f(oy*inv)*radius)
    return ox,oy
def _lp1_point(L,k,radius,ox,oy):
    px,py,dx,dy=L[k]
    dp=f(f(px*dx)+f(py*dy))
    disc=f(f(f(dp*dp)+f(radius*radius))-f(f(px*px)+f(py*py)))
    if disc<0: return None
    root=np.sqrt(disc); tlo=f(-dp-root); thi=f(-dp+root)
    for i in range(k):
        qx,qy,ex,ey=L[i]
        den=f(f(dx*ey)-f(dy*ex)); num=f(f(ex*f(py-qy))-f(ey*f(px-qx)))
        if abs(den)<=EPS:
            if num<0: return None
            continue
        t=f(num/den)
        if den>=0: thi = t if t<thi else thi
        else: tlo = t if tlo<t else tlo
        if tlo>thi: return None
    t=f(f(dx*f(ox-px))+f(dy*f(oy-py)))
    if t<tlo: t=tlo
    elif t>thi: t=thi
    return f(px+f(t*dx)), f(py+f(t*dy))
def seq(L,n,radius,ox,oy):
    rx,ry=start(radius,ox,oy)
    for i in range(n):
        px,py,dx,dy=L[i]
        if f(f(dx*f(py-ry))-f(dy*f(px-rx)))>0:
            r=_lp1_point(L,i,radius,ox,oy)
            if r is None: return rx,ry,i
            rx,ry=r
    return rx,ry,n
def coop(L,n,radius,ox,oy,MAXL=5):
    rx,ry=start(radius,ox,oy); cursor=0; fail=n
    with np.errstate(all='ignore'):
      while True:
        viol=[l>=cursor and l<n and f(f(L[l][2]*f(L[l][1]-ry))-f(L[l][3]*f(L[l][0]-rx)))>0 for l in range(MAXL)]
        if not any(viol): break
        i=viol.index(True)
        px,py,dx,dy=L[i]
        chi=[f(np.inf)]*MAXL; clo=[f(-np.inf)]*MAXL; bad=False
        for l in range(MAXL):
            mx,my,mz,mw=L[l] if l<n else (f(0),f(0),f(0),f(0))
            den=f(f(dx*mw)-f(dy*mz)); num=f(f(mz*f(py-my))-f(mw*f(px-mx)))
            par=abs(den)<=EPS; t=f(num/den) if den!=0 else f(np.nan)
            mine=l<i
            if mine and par and num<0: bad=True
            if mine and not par and den>=0: chi[l]=t
            if mine and not par and not den>=0: clo[l]=t
        dp=f(f(px*dx)+f(py*dy)); disc=f(f(f(dp*dp)+f(radius*radius))-f(f(px*px)+f(py*py)))
        ok=(not disc<0) and not bad
        root=np.sqrt(disc); tlo=f(-dp-root); thi=f(-dp+root)
        for j in range(MAXL-1):
            thi = chi[j] if chi[j]<thi else thi
            tlo = clo[j] if tlo<clo[j] else tlo
        ok = ok and not (tlo>thi)
        tt=f(f(dx*f(ox-px))+f(dy*f(oy-py)))
        tt = tlo if tt<tlo else (thi if tt>thi else tt)
        if ok: rx,ry=f(px+f(tt*dx)),f(py+f(tt*dy)); cursor=i+1
        else: fail=i; cursor=n
    return rx,ry,fail

def lp1(L,k,radius,ox,oy,dir_opt):
    px,py,dx,dy=L[k]
    dp=f(f(px*dx)+f(py*dy))
    disc=f(f(f(dp*dp)+f(radius*radius))-f(f(px*px)+f(py*py)))
    if disc<0: return None
    root=np.sqrt(disc); tlo=f(-dp-root); thi=f(-dp+root)
    for i in range(k):
        qx,qy,ex,ey=L[i]
        den=f(f(dx*ey)-f(dy*ex)); num=f(f(ex*f(py-qy))-f(ey*f(px-qx)))
        if abs(den)<=EPS:
            if num<0: return None
            continue
        t=f(num/den)
        if den>=0: thi = t if t<thi else thi
        else: tlo = t if tlo<t else tlo
        if tlo>thi: return None
    if dir_opt:
        t = thi if f(f(ox*dx)+f(oy*dy))>0 else tlo
    else:
        t=f(f(dx*f(ox-px))+f(dy*f(oy-py)))
        if t<tlo: t=tlo
        elif t>thi: t=thi
    return f(px+f(t*dx)), f(py+f(t*dy))
def lp2(L,n,radius,ox,oy,dir_opt,rx,ry):
    if dir_opt: rx,ry=f(ox*radius),f(oy*radius)
    for i in range(n):
        px,py,dx,dy=L[i]
        if f(f(dx*f(py-ry))-f(dy*f(px-rx)))>0:
            r=lp1(L,i,radius,ox,oy,dir_opt)
            if r is None: return rx,ry,i
            rx,ry=r
    return rx,ry,n
def lp3_seq(L,n,begin,radius,rx,ry):
    distance=f(0)
    for i in range(begin,n):
        pix,piy,dix,diy=L[i]
        if f(f(dix*f(piy-ry))-f(diy*f(pix-rx)))>distance:
            P=[]
            for j in range(i):
                pjx,pjy,djx,djy=L[j]
                d=f(f(dix*djy)-f(diy*djx))
                if abs(d)<=EPS:
                    if f(f(dix*djx)+f(diy*djy))>0: continue
                    qx=f(f(0.5)*f(pix+pjx)); qy=f(f(0.5)*f(piy+pjy))
                else:
                    t=f(f(f(djx*f(piy-pjy))-f(djy*f(pix-pjx)))/d)
                    qx=f(pix+f(t*dix)); qy=f(piy+f(t*diy))
                ex=f(djx-dix); ey=f(djy-diy)
                inv=f(f(1.0)/np.sqrt(f(f(ex*ex)+f(ey*ey))))
                P.append((qx,qy,f(ex*inv),f(ey*inv)))
            kx,ky=rx,ry
            rx2,ry2,fl=lp2(P,len(P),radius,f(-diy),dix,True,rx,ry)
            if fl<len(P): rx,ry=kx,ky
            else: rx,ry=rx2,ry2
            distance=f(f(dix*f(piy-ry))-f(diy*f(pix-rx)))
    return rx,ry
def lp3_coop(L,n,begin,radius,rx,ry,MAXL=5):
    distance=f(0); icur=begin
    with np.errstate(all='ignore'):
      while True:
        cond=[l>=icur and l<n and f(f(L[l][2]*f(L[l][1]-ry))-f(L[l][3]*f(L[l][0]-rx)))>distance for l in range(MAXL)]
        if not any(cond): break
        i=cond.index(True)
        pix,piy,dix,diy=L[i]
        # projection per lane j<i
        P=[None]*MAXL; valid=[False]*MAXL
        for j in range(MAXL):
            if j>=i or j>=n: continue
            pjx,pjy,djx,djy=L[j]
            d=f(f(dix*djy)-f(diy*djx))
            par=abs(d)<=EPS
            if par and f(f(dix*djx)+f(diy*djy))>0: continue
            t=f(f(f(djx*f(piy-pjy))-f(djy*f(pix-pjx)))/d) if d!=0 else f(np.nan)
            qx=f(f(0.5)*f(pix+pjx)) if par else f(pix+f(t*dix)); qy=f(f(0.5)*f(piy+pjy)) if par else f(piy+f(t*diy))
            ex=f(djx-dix); ey=f(djy-diy); inv=f(f(1.0)/np.sqrt(f(f(ex*ex)+f(ey*ey))))
            P[j]=(qx,qy,f(ex*inv),f(ey*inv)); valid[j]=True
        ox,oy=f(-diy),dix
        r2x,r2y=f(ox*radius),f(oy*radius); cur2=0; failed=False
        while True:
            viol=[valid[l] and l>=cur2 and l<i and f(f(P[l][2]*f(P[l][1]-r2y))-f(P[l][3]*f(P[l][0]-r2x)))>0 for l in range(MAXL)]
            if not any(viol): break
            k=viol.index(True)
            px,py,dx,dy=P[k]
            chi=[INF]*MAXL; clo=[-INF]*MAXL; bad=False
            for l in range(MAXL):
                if not (valid[l] and l<k): continue
                mx,my,mz,mw=P[l]
                den=f(f(dx*mw)-f(dy*mz)); num=f(f(mz*f(py-my))-f(mw*f(px-mx)))
                par=abs(den)<=EPS; t=f(num/den) if den!=0 else f(np.nan)
                if par and num<0: bad=True
                if not par and den>=0: chi[l]=t
                if not par and not den>=0: clo[l]=t
            dp=f(f(px*dx)+f(py*dy)); disc=f(f(f(dp*dp)+f(radius*radius))-f(f(px*px)+f(py*py)))
            ok=(not disc<0) and not bad
            root=np.sqrt(disc); tlo=f(-dp-root); thi=f(-dp+root)
            for j in range(MAXL-1):
                thi = chi[j] if chi[j]<thi else thi
                tlo = clo[j] if tlo<clo[j] else tlo
            ok = ok and not (tlo>thi)
            t = thi if f(f(ox*dx)+f(oy*dy))>0 else tlo
            if ok: r2x,r2y=f(px+f(t*dx)),f(py+f(t*dy)); cur2=k+1
            else: failed=True; cur2=i
        if not failed: rx,ry=r2x,r2y
        distance=f(f(dix*f(piy-ry))-f(diy*f(pix-rx)))
        icur=i+1
    return rx,ry


def _random_program(rng, maxl):
    lines = []
    for _ in range(maxl):
        ang = rng.uniform(0, 2 * np.pi)
        if rng.rand() < 0.2 and lines:  # parallel / anti-parallel to the previous line
            ang = np.arctan2(float(lines[-1][3]), float(lines[-1][2])) + (np.pi if rng.rand() < 0.5 else 0)
        sc = rng.choice([0.05, 0.3, 1.0, 2.0])
        lines.append((f(rng.normal() * sc), f(rng.normal() * sc), f(np.cos(ang)), f(np.sin(ang))))
    return lines, f(rng.choice([0.5, 1.0, 1.5])), f(rng.normal()), f(rng.normal())


def _same(a, b):
    return np.array(a, dtype=f).tobytes() == np.array(b, dtype=f).tobytes()


def test_cooperative_planar_program_equals_sequential():
    rng = np.random.RandomState(0)
    infeasible = 0
    for trial in range(3000):
        maxl = 5 if trial % 2 else 10
        lines, radius, ox, oy = _random_program(rng, maxl)
        n = rng.randint(0, maxl + 1)
        a, b = seq(lines, n, radius, ox, oy), coop(lines, n, radius, ox, oy, maxl)
        assert a[2] == b[2] and _same(a[:2], b[:2]), trial
        infeasible += a[2] < n
    assert infeasible > 500


def test_cooperative_fallback_equals_sequential():
    rng = np.random.RandomState(1)
    cases = 0
    for trial in range(4000):
        maxl = 5 if trial % 2 else 10
        lines, radius, ox, oy = _random_program(rng, maxl)
        n = rng.randint(1, maxl + 1)
        sx, sy = start(radius, ox, oy)
        rx, ry, fail = lp2(lines, n, radius, ox, oy, False, sx, sy)
        if fail < n:
            assert _same(lp3_seq(lines, n, fail, radius, rx, ry), lp3_coop(lines, n, fail, radius, rx, ry, maxl)), trial
            cases += 1
    assert cases > 1000

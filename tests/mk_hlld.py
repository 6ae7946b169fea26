"""Independent HLLD from the equations of Miyoshi & Kusano, JCP 208, 315 (2005) -- used only
to cross-check the oracle's restatement (tests/test_oracle_pins.py)."""
import numpy as np


def hlld_mk(g, wl, wr, bx):
    def prim2(w):
        d,vx,vy,vz,e,by,bz = w
        p=(g-1)*e
        pt = p+0.5*(bx*bx+by*by+bz*bz)
        E = p/(g-1)+0.5*d*(vx*vx+vy*vy+vz*vz)+0.5*(bx*bx+by*by+bz*bz)
        U=np.array([d,d*vx,d*vy,d*vz,E,by,bz])
        F=np.array([d*vx, d*vx*vx+pt-bx*bx, d*vy*vx-bx*by, d*vz*vx-bx*bz,
                    (E+pt)*vx-bx*(vx*bx+vy*by+vz*bz), by*vx-bx*vy, bz*vx-bx*vz])
        a2=g*p; ct2=by*by+bz*bz; q=bx*bx+ct2+a2; t=bx*bx+ct2-a2
        cf=np.sqrt(0.5*(q+np.sqrt(t*t+4*a2*ct2))/d)
        return d,vx,vy,vz,p,pt,E,by,bz,U,F,cf
    dl,ul,vl,wl_,pl,ptl,El,byl,bzl,UL,FL,cfl=prim2(wl)
    dr,ur,vr,wr_,pr,ptr,Er,byr,bzr,UR,FR,cfr=prim2(wr)
    SL=min(ul-cfl,ur-cfr); SR=max(ul+cfl,ur+cfr)
    SM=((SR-ur)*dr*ur-(SL-ul)*dl*ul-ptr+ptl)/((SR-ur)*dr-(SL-ul)*dl)   # eq 38
    pts = ((SR-ur)*dr*ptl-(SL-ul)*dl*ptr+dl*dr*(SR-ur)*(SL-ul)*(ur-ul))/((SR-ur)*dr-(SL-ul)*dl) # eq 41
    def star(S,d,u,v,w,pt,E,by,bz):
        ds=d*(S-u)/(S-SM)  # 43
        den=d*(S-u)*(S-SM)-bx*bx
        if abs(den)<1e-4*pts:
            vs,ws,bys,bzs=v,w,by,bz
        else:
            vs=v-bx*by*(SM-u)/den; ws=w-bx*bz*(SM-u)/den  # 44,46
            bys=by*(d*(S-u)**2-bx*bx)/den; bzs=bz*(d*(S-u)**2-bx*bx)/den # 45 47
        vb=u*bx+v*by+w*bz; vbs=SM*bx+vs*bys+ws*bzs
        Es=((S-u)*E-pt*u+pts*SM+bx*(vb-vbs))/(S-SM)  # 48
        return ds,vs,ws,bys,bzs,Es
    dls,vls,wls,byls,bzls,Els=star(SL,dl,ul,vl,wl_,ptl,El,byl,bzl)
    drs,vrs,wrs,byrs,bzrs,Ers=star(SR,dr,ur,vr,wr_,ptr,Er,byr,bzr)
    ULs=np.array([dls,dls*SM,dls*vls,dls*wls,Els,byls,bzls])
    URs=np.array([drs,drs*SM,drs*vrs,drs*wrs,Ers,byrs,bzrs])
    sdl=np.sqrt(dls); sdr=np.sqrt(drs)
    SLs=SM-abs(bx)/sdl; SRs=SM+abs(bx)/sdr  # 51
    if 0.5*bx*bx<1e-4*pts:
        ULss,URss=ULs.copy(),URs.copy()
    else:
        sg=1.0 if bx>0 else -1.0
        vss=(sdl*vls+sdr*vrs+(byrs-byls)*sg)/(sdl+sdr)  #59
        wss=(sdl*wls+sdr*wrs+(bzrs-bzls)*sg)/(sdl+sdr)  #60
        byss=(sdl*byrs+sdr*byls+sdl*sdr*(vrs-vls)*sg)/(sdl+sdr) #61
        bzss=(sdl*bzrs+sdr*bzls+sdl*sdr*(wrs-wls)*sg)/(sdl+sdr) #62
        vbss=SM*bx+vss*byss+wss*bzss
        Elss=Els-sdl*((SM*bx+vls*byls+wls*bzls)-vbss)*sg  #63
        Erss=Ers+sdr*((SM*bx+vrs*byrs+wrs*bzrs)-vbss)*sg
        ULss=np.array([dls,dls*SM,dls*vss,dls*wss,Elss,byss,bzss])
        URss=np.array([drs,drs*SM,drs*vss,drs*wss,Erss,byss,bzss])
    if SL>=0: return FL
    if SR<=0: return FR
    if SLs>=0: return FL+SL*(ULs-UL)
    if SM>=0: return FL+SLs*ULss-(SLs-SL)*ULs-SL*UL
    if SRs>0: return FR+SRs*URss-(SRs-SR)*URs-SR*UR
    return FR+SR*(URs-UR)

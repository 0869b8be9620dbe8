import sys, time; sys.path.insert(0,"tests"); sys.path.insert(0,".")
import numpy as np, torch, helpers as H, gpumd_amd
dev=torch.device("cuda",0)
model=gpumd_amd.Model(H.golden("PbTe","nep.txt"))
h,typ,x=H.pbte_supercell((10,10,10),rattle=0.02,seed=5)
n=len(typ)
mass=np.where(typ==0,H.MASS["Te"],H.MASS["Pb"]).astype(np.float64)
for T0 in (1000.0, 2500.0):
    vel=H.maxwell_velocities(mass,T0,seed=3)
    eng=gpumd_amd.NEP(model,n)
    t=[torch.from_numpy(a).to(dev) for a in (typ,x.copy(),vel,mass)]
    pe,f,w=(torch.zeros(k*n,dtype=torch.float64,device=dev) for k in (1,3,9))
    eng.force_compute(h,t[0],t[1],pe,f,w)
    try:
        th=eng.run_nve(h,t[0],t[3],1.0/H.TIME_UNIT,1500,t[1],t[2],pe,f,w,thermo_every=500)
        st=eng.stats(True)
        e=(th[:,1]+1.5*n*H.K_B*th[:,0])/n
        print("T0",T0,"ok: T_end %.0f K, rebuilds %d, max nn skin/rad/ang %d %d %d, E drift %.2e eV/atom"%(th[-1,0],st.num_rebuild,st.max_nn_skin,st.max_nn_radial,st.max_nn_angular,abs(e[-1]-e[0])))
    except Exception as ex:
        print("T0",T0,"FAILED:",ex)

import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.sparse as sp
import juliagrid.jl_amd as jg
from oracle import oracle
from test_oracle_se import se_case14
from test_se_gpu import _mirror, _system_like
from conftest import load_case
t,osys,vm,va = se_case14(oracle)
tab=oracle.MeterTable()
oracle.add_from_power_flow(tab,osys,vm,va,"pmu",bus=True,frm=False,to=False,variance=1.0,polar=True)
oracle.add_from_power_flow(tab,osys,vm,va,"ammeter",variance=1e-4,square=True)
s=_system_like(jg,t,osys)
an=jg.gaussNewton(_mirror(jg,s,tab)); gn=oracle.OracleGN(osys,tab)
for it in range(30):
    a=jg.incrementSE_(an); b=gn.increment()
    print(it,"dev %.3e oracle %.3e inc diff %.2e"%(a,b,np.abs(an.increment-gn.vectors()["increment"]).max()))
    jg.solveSE_(an); gn.solve()
print("==== case1354")
t=load_case("case1354pegase"); s=jg.powerSystem(t); pf=jg.newtonRaphson(s); jg.powerFlow_(pf,tolerance=1e-12)
mon=jg.measurement(s); jg.addVoltmeter_(mon,pf); jg.addAmmeter_(mon,pf); jg.addWattmeter_(mon,pf); jg.addVarmeter_(mon,pf); jg.addPmu_(mon,pf)
q=jg.exactQuantities(s,pf.voltage.magnitude,pf.voltage.angle)
print("min current", q.fromMagnitude[s.branch.layout.status==1].min(), q.toMagnitude[s.branch.layout.status==1].min())
an=jg.gaussNewton(mon)
print("wdiag finite", np.isfinite(an.method._wdiag).all(), an.method._wdiag.max())
for it in range(12):
    try:
        a=jg.incrementSE_(an)
    except Exception as e:
        print("ERR",e); break
    print(it,"dev %.3e"%a, "dV %.2e"%np.abs(an.voltage.magnitude-pf.voltage.magnitude).max())
    jg.solveSE_(an)

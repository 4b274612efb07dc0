import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import torch
from conftest import load_case
from oracle import oracle
import juliagrid.jl_amd as jg
from test_se_gpu import _mirror, _system_like
from test_se_scale_gpu import _oracle_system
t = load_case("case1354pegase")
osys, vm, va = _oracle_system(oracle, t)
fams = sys.argv[1:] or ["voltmeter","ammeter","ammeter2","wattmeter","varmeter","pmupolar","pmupolarsq","pmu","pmucorr"]
tab = oracle.MeterTable()
for f in fams:
    if f=="voltmeter": oracle.add_from_power_flow(tab, osys, vm, va, "voltmeter")
    if f=="ammeter": oracle.add_from_power_flow(tab, osys, vm, va, "ammeter", variance=1e-4)
    if f=="ammeter2": oracle.add_from_power_flow(tab, osys, vm, va, "ammeter", variance=1e-4, square=True)
    if f=="wattmeter": oracle.add_from_power_flow(tab, osys, vm, va, "wattmeter")
    if f=="varmeter": oracle.add_from_power_flow(tab, osys, vm, va, "varmeter")
    if f=="pmupolar": oracle.add_from_power_flow(tab, osys, vm, va, "pmu", polar=True)
    if f=="pmupolarsq": oracle.add_from_power_flow(tab, osys, vm, va, "pmu", bus=False, polar=True, square=True)
    if f=="pmu": oracle.add_from_power_flow(tab, osys, vm, va, "pmu")
    if f=="pmucorr": oracle.add_from_power_flow(tab, osys, vm, va, "pmu", correlated=True)
tab.rows = [r for r in tab.rows if not (r[0] in (2, 5) and r[1] != 0 and r[3] < 5e-2)]
s = _system_like(jg, t, osys)
v0, a0 = np.asarray(t["bus_vm"], dtype=float), np.asarray(t["bus_va"], dtype=float)
an = jg.gaussNewton(_mirror(jg, s, tab))
gn = oracle.OracleGN(osys, tab, v0, a0)
an.setVoltage(v0, a0)
mo = gn.increment(); v = gn.vectors()
print(fams, "oracle inc max", mo, "H finite", np.isfinite(v["jacobian"]).all(), "Hmax", np.abs(v["jacobian"]).max(), "wdiag range", gn.wdiag.min(), gn.wdiag.max())
try:
    mx = jg.incrementSE_(an); print("device ok", mx, np.abs(an.increment - v["increment"]).max())
except Exception as e:
    print("device failed:", e)
    jg._lib.lib().jg_gn_evaluate(an._h)
    H = an.jacobian.nzval
    print("device H finite", np.isfinite(H).all(), "diff vs oracle", np.abs(H - v["jacobian"]).max())
# per type code: largest deviation of H (device vs oracle) at the start point, and of the residual
jg._lib.lib().jg_gn_evaluate(an._h)
H = an.jacobian
rows = H.rowval - 1
d = np.abs(H.nzval - v["jacobian"])
typ = gn.type[rows]
for c in np.unique(gn.type):
    sel = typ == c
    if sel.any():
        k = np.argmax(d * sel)
        r = rows[k]
        print("type %2d: rows %6d  max |dH| %.3e at |H| %.3e (row max |H| %.3e)  max |dres| %.3e" % (c, int((gn.type == c).sum()), d[sel].max(), abs(v["jacobian"][k]), np.abs(v["jacobian"][rows == r]).max(), np.abs(an.residual - v["residual"])[gn.type == c].max()))

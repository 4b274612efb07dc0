import sys,os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, juliagrid.jl_amd as jg
from conftest import load_case
s=jg.powerSystem(load_case("case_ACTIVSg10k")); labels=jg.outageList(s,512)
an=jg.contingencyAnalysis(s,labels)
jg.mismatch_(an)
print("dbg",os.environ.get("JG_DBG"),"assembly ms",an.time_kernel(0,30))

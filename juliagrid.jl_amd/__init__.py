"""juliagrid.jl_amd -- MI355X-native Newton-Raphson power flow / Gauss-Newton state estimation.

Host-side mirror of the JuliaGrid interface for the hot path only; all numerics run
in libjgrid_hip.so (hand-written HIP for gfx950) through the C ABI of include/jgrid.h.
"""
from .system import PowerSystem, CscMatrix, powerSystem, acModel_          # noqa: F401
from .system import addBranch_ as addBranchSystem_, dropZeros_ as dropZerosSystem_   # noqa: F401
from .system import (updateBranch_ as updateBranchSystem_, updateBus_ as updateBusSystem_,   # noqa: F401
                     updateGenerator_ as updateGeneratorSystem_)
from .powerflow import (AcPowerFlow, newtonRaphson, fastNewtonRaphsonBX, fastNewtonRaphsonXB, mismatch_, solve_, powerFlow_, setInitialPoint_, setRefinement_,   # noqa: F401
                        updateBranch_, updateBus_, updateGenerator_, addBranch_, dropZeros_, setOutage_, setOutages_, setInjection_, outagePatch, fastOutagePatch, initializeACPowerFlow, power_, current_, screenSummary_, reactiveLimit_, adjustAngle_,
                        BaseCase, startFromBase_, setFirstIteration_, firstIterationCounts)
from .contingency import bridges, outageList, shard, deviceBatching, recommendedLanes, contingencyAnalysis, gatherResults, gatherResultsDevice, unpackResults, ContingencyPipeline   # noqa: F401
from .measurement import (Measurement, measurement, ems, addVoltmeter_, addAmmeter_, addWattmeter_, addVarmeter_,   # noqa: F401
                          addPmu_, exactQuantities)
from .stateestimation import (WlsMethod, Normal, LU, KLU, QR, LDLt, LL, Orthogonal, PetersWilkinson,   # noqa: F401
                              AcStateEstimation, PmuStateEstimation, pmuStateEstimation, gaussNewton, increment_ as incrementSE_, solve_ as solveSE_,   # noqa: F401
                              stateEstimation_, setNoise_, drawNoise_, measurementDevice, residualTest_, normalizedResidual, chiTest,
                              updateVoltmeter_, updateAmmeter_, updateWattmeter_, updateVarmeter_, updatePmu_)
from .montecarlo import MonteCarloPipeline, gatherEstimates, gatherEstimatesDevice, unpackEstimates   # noqa: F401
from .synthetic import pegaseShaped, case9241synth                          # noqa: F401
from . import powerflow, stateestimation   # noqa: F401
from . import _lib                                                           # noqa: F401

__all__ = [
    "PowerSystem", "CscMatrix", "powerSystem", "acModel_", "updateBranchSystem_", "updateBusSystem_", "updateGeneratorSystem_", "updateBus_", "updateGenerator_", "AcPowerFlow", "newtonRaphson", "fastNewtonRaphsonBX", "fastNewtonRaphsonXB",
    "mismatch_", "solve_", "powerFlow_", "setInitialPoint_", "setRefinement_", "updateBranch_", "setOutage_", "setInjection_",
    "Measurement", "measurement", "ems", "addVoltmeter_", "addAmmeter_", "addWattmeter_", "addVarmeter_", "addPmu_",
    "exactQuantities", "AcStateEstimation", "PmuStateEstimation", "pmuStateEstimation", "gaussNewton", "incrementSE_", "solveSE_", "stateEstimation_", "setNoise_", "drawNoise_", "measurementDevice", "residualTest_", "normalizedResidual", "chiTest",
    "updateVoltmeter_", "updateAmmeter_", "updateWattmeter_", "updateVarmeter_", "updatePmu_",
    "outagePatch", "fastOutagePatch", "initializeACPowerFlow", "bridges", "outageList", "shard", "deviceBatching", "recommendedLanes", "contingencyAnalysis", "gatherResults", "gatherResultsDevice", "unpackResults",
    "WlsMethod", "Normal", "LU", "KLU", "QR", "LDLt", "LL", "Orthogonal", "PetersWilkinson",
    "addBranch_", "dropZeros_", "addBranchSystem_", "dropZerosSystem_", "pegaseShaped", "case9241synth", "ContingencyPipeline", "MonteCarloPipeline", "gatherEstimates", "gatherEstimatesDevice", "unpackEstimates", "setOutages_", "power_", "current_", "screenSummary_", "reactiveLimit_", "adjustAngle_",
    "BaseCase", "startFromBase_", "setFirstIteration_", "firstIterationCounts",
]

"""The seeded PEGASE-shaped 9 241-bus grid that stands in for case9241pegase:
shape statistics, determinism, connectivity, and the acceptance rule (the CPU oracle converges from the flat start
in <= 10 Newton-Raphson iterations)."""
import numpy as np

from conftest import load_case


def test_shape_and_determinism(jg):
    from juliagrid.jl_amd.synthetic import case9241synth
    t, t2 = load_case("case9241synth"), case9241synth()
    assert all(np.array_equal(t[k], t2[k]) for k in t)                       # PCG64(seed): bit-reproducible
    n, nb, ng = t["bus_type"].size, t["br_from"].size, t["gen_bus"].size
    assert (n, nb, ng) == (9241, 16049, 1445)
    pairs = np.minimum(t["br_from"], t["br_to"]) * (n + 1) + np.maximum(t["br_from"], t["br_to"])
    dup = 1.0 - np.unique(pairs).size / nb
    assert 0.10 <= dup <= 0.16                                              # case1354pegase: 14 % parallel circuits
    assert n + 2 * np.unique(pairs).size <= 41339                           # nnz(Y) bound of SURVEY 8
    assert np.all(t["br_from"] != t["br_to"])
    assert np.bincount(t["bus_type"])[3] == 1 and abs(np.mean(t["bus_type"] == 2) - 0.156) < 0.002
    tr = (t["br_tap"] != 1.0) | (t["br_shift"] != 0.0)
    assert 0.09 <= tr.mean() <= 0.16                                       # case1354pegase: 11.8 % (+ their parallel twins here)
    deg = np.bincount(np.concatenate([t["br_from"], t["br_to"]]), minlength=n + 1)[1:]
    assert deg.min() >= 1 and 0.15 <= np.mean(deg == 1) <= 0.45           # radial leaves fed by a single circuit
    s = jg.powerSystem(t)
    assert len(jg.bridges(s)) > 2000                                         # the radial leaves hang on bridges
    jg.acModel_(s)
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    Y = s.model.ac.nodalMatrix
    A = sp.csc_matrix((np.ones(Y.nnz), Y.rowval - 1, Y.colptr - 1), shape=(n, n))
    assert connected_components(A, directed=False)[0] == 1
    assert s.model.ac.nodalMatrix.nnz == n + 2 * np.unique(pairs).size


def test_oracle_converges_from_flat_start(oracle):
    t = load_case("case9241synth")
    o = oracle.OracleNR(oracle.OracleSystem(t))
    assert o.power_flow(iteration=10, tolerance=1e-8) == 0
    assert o.iteration <= 10
    assert 16900 <= o.dim <= 17100 and 1.1e5 <= o.nnzJ <= 1.4e5              # dimJ ~ 17 0xx, nnz(J) ~ 1.3e5
    vm, va = o.voltage()
    assert vm.min() > 0.9 and vm.max() < 1.2


def test_factor_structure_is_grid_like(jg):
    """The meshed backbone gives fronts of realistic width (ACTIVSg10k: 43) - not a tree, not a random graph."""
    s = jg.powerSystem("case9241synth")
    jg.acModel_(s)
    Y = s.model.ac.nodalMatrix
    plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1)
    front = np.diff(plan.get("u_ptr")).max()
    assert 15 <= front <= 80
    assert plan.get("e_row").size <= 3 * Y.nnz


def test_tiled_grid_shape_and_determinism(oracle):
    """tiledGrid: 7 tied instances of case_ACTIVSg10k (stands in for the reference's 70 000-bus dataset): table sizes, one slack, the
    ties, seeded jitter, and the acceptance rule of every stand-in -- the oracle's Newton-Raphson converges from the stored start."""
    from conftest import load_case
    from juliagrid.jl_amd.synthetic import tiledGrid
    t = load_case("case_ACTIVSg10k")
    o = oracle.OracleNR(oracle.OracleSystem(t))
    assert o.power_flow() == 0
    vm, va = o.voltage()
    slack = int(np.flatnonzero(t["bus_type"] == 3)[0])
    p_slack = oracle.exact_quantities(oracle.OracleSystem(t), vm, va)[1][slack, 0] + t["bus_pd"][slack]
    a = tiledGrid(t, 7, slack_active=p_slack)
    b = tiledGrid(t, 7, slack_active=p_slack)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    n, nb, ng = t["bus_type"].size, t["br_from"].size, t["gen_bus"].size
    assert a["bus_type"].size == 7 * n and a["br_from"].size == 7 * nb + 6 and a["gen_bus"].size == 7 * ng
    assert int(np.sum(a["bus_type"] == 3)) == 1 and a["br_from"].max() <= 7 * n and a["gen_bus"].max() <= 7 * n
    assert not np.array_equal(a["br_x"][:nb], a["br_x"][nb:2 * nb])            # instances are not bit-identical
    big = oracle.OracleNR(oracle.OracleSystem(a))
    assert big.power_flow(iteration=20, tolerance=1e-8) == 0 and big.iteration <= 10

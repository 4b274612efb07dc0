"""Host-side mirror (juliagrid.jl_amd.system / powerflow bookkeeping) against the oracle. CPU only."""
import numpy as np
import pytest

from conftest import ROOT, load_case

CASES = ["case14", "case14test", "case30test", "case118", "case300", "case1354pegase", "case_ACTIVSg10k"]


@pytest.mark.parametrize("name", CASES)
def test_ac_model_matches_oracle(jg, oracle, name):
    t = load_case(name)
    s = jg.powerSystem(t)
    jg.acModel_(s)
    o = oracle.OracleSystem(t)
    Y, YT = s.model.ac.nodalMatrix, s.model.ac.nodalMatrixTranspose
    assert np.array_equal(Y.colptr, o.colptr) and np.array_equal(Y.rowval, o.rowval)      # bit-exact pattern
    scale = np.abs(o.ybus).max()
    assert np.abs(Y.nzval - o.ybus).max() <= 1e-14 * scale
    assert np.abs(YT.nzval - (o.ytre + 1j * o.ytim)).max() <= 1e-14 * scale
    tp = o.twoport.reshape(-1, 5, 2)
    for k, arr in enumerate([s.model.ac.admittance, s.model.ac.nodalFromFrom, s.model.ac.nodalFromTo,
                             s.model.ac.nodalToTo, s.model.ac.nodalToFrom]):
        assert np.abs(arr - (tp[:, k, 0] + 1j * tp[:, k, 1])).max() <= 1e-14 * scale


def test_out_of_service_branches_are_stored_zeros(jg):
    """model.jl:70-71: status 0 branches still insert (from,to),(to,from) entries with value 0."""
    t = load_case("case14test")
    s = jg.powerSystem(t)
    jg.acModel_(s)
    off = np.flatnonzero(t["br_status"] == 0)
    assert off.size > 0
    Y = s.model.ac.nodalMatrix
    for k in off:
        f, to = int(t["br_from"][k]), int(t["br_to"][k])
        Y.position(f, to)
        Y.position(to, f)                      # raises if the entry is not stored


@pytest.mark.parametrize("name", CASES)
def test_bus_type_normalisation_matches_oracle(jg, oracle, name):
    t = load_case(name)
    s = jg.powerSystem(t)
    jg.acModel_(s)
    vm, va = jg.initializeACPowerFlow(s)
    o = oracle.OracleNR(oracle.OracleSystem(t))
    assert np.array_equal(s.bus.layout.type, o.type)
    assert s.bus.layout.slack == o.slack
    assert np.array_equal(vm, o.vm) and np.array_equal(va, o.va)


def test_update_branch_outage_and_reclose(jg):
    """branch.jl:344-350 / 381-386: outage keeps the pattern (stored zeros); re-close restores values
    to within an ulp."""
    s = jg.powerSystem(load_case("case14"))
    jg.acModel_(s)
    Y0 = s.model.ac.nodalMatrix.nzval.copy()
    nnz0 = s.model.ac.nodalMatrix.nnz
    jg.updateBranchSystem_(s, 5, status=0)
    assert s.model.ac.nodalMatrix.nnz == nnz0
    f, t = int(s.branch.layout.from_[4]), int(s.branch.layout.to[4])
    assert s.model.ac.nodalMatrix.nzval[s.model.ac.nodalMatrix.position(f, t)] == 0
    assert s.model.revision.topology == 1
    jg.updateBranchSystem_(s, 5, status=1)
    assert np.abs(s.model.ac.nodalMatrix.nzval - Y0).max() < 1e-13
    # transpose copy kept in sync
    from importlib import import_module
    sysmod = import_module("juliagrid.jl_amd.system")
    assert np.array_equal(sysmod._transpose_values(s.model.ac.nodalMatrix), s.model.ac.nodalMatrixTranspose.nzval)


def test_matpower_reader_matches_fixture(jg, tmp_path):
    """The .m reader follows load.jl:341-619 unit conventions: same tables as the committed fixture."""
    t = load_case("case14test")
    # write a tiny MATPOWER file from the fixture (MW / degrees) and read it back
    base = 100.0
    lines = ["function mpc = x", "mpc.baseMVA = 100;", "mpc.bus = ["]
    for i in range(t["bus_type"].size):
        lines.append(f" {t['bus_label'][i]} {t['bus_type'][i]} {t['bus_pd'][i]*base!r} {t['bus_qd'][i]*base!r} "
                     f"{t['bus_gs'][i]*base!r} {t['bus_bs'][i]*base!r} 1 {t['bus_vm'][i]!r} {np.rad2deg(t['bus_va'][i])!r} 138 1 1.06 0.94;")
    lines += ["];", "mpc.gen = ["]
    for k in range(t["gen_bus"].size):
        lines.append(f" {t['bus_label'][t['gen_bus'][k]-1]} {t['gen_pg'][k]*base!r} {t['gen_qg'][k]*base!r} 10 -10 {t['gen_vg'][k]!r} 100 {t['gen_status'][k]} 100 0;")
    lines += ["];", "mpc.branch = ["]
    for k in range(t["br_from"].size):
        lines.append(f" {t['bus_label'][t['br_from'][k]-1]} {t['bus_label'][t['br_to'][k]-1]} {t['br_r'][k]!r} {t['br_x'][k]!r} {t['br_b'][k]!r} 0 0 0 "
                     f"{t['br_tap'][k]!r} {np.rad2deg(t['br_shift'][k])!r} {t['br_status'][k]} -360 360;")
    lines += ["];"]
    p = tmp_path / "x.m"
    p.write_text("\n".join(lines).replace("np.float64(", "").replace(")", ""))
    s = jg.powerSystem(str(p))
    assert s.bus.number == 14 and s.branch.number == 20
    assert np.array_equal(s.bus.layout.type, t["bus_type"])
    assert np.array_equal(s.branch.layout.from_, t["br_from"])
    assert np.abs(s.bus.demand.active - t["bus_pd"]).max() < 1e-15
    assert np.abs(s.branch.parameter.reactance - t["br_x"]).max() == 0


def test_hdf5_reader_matches_the_fixtures(jg):
    """N1: the pure-Python reader of the reference's HDF5 case layout (compact and dense groups, contiguous datasets,
    one-element datasets broadcast) gives bit-identical tables to the converter that went through h5dump."""
    import glob
    import os
    from juliagrid.jl_amd.hdf5 import H5File, case_tables
    here = os.path.join(ROOT, "tests", "golden", "h5")
    files = {os.path.basename(p)[:-3]: p for p in glob.glob(os.path.join(here, "case*.h5"))}
    assert set(files) == {"case14", "case_ieee30"}
    for p in glob.glob("/root/reference/docs/src/examples/cases/hdf5/*.h5"):      # build container only: every shipped case
        files.setdefault(os.path.basename(p)[:-3], p)
    for name, p in files.items():
        t = case_tables(p)
        ref = load_case(name)
        for k, v in t.items():
            assert np.array_equal(np.asarray(v).reshape(-1), np.asarray(ref[k]).reshape(-1)), (name, k)
        f = H5File(p)
        assert len(f.datasets()) == 56 and not f.skipped                         # every group enumerated, dense ones included
        assert f.read("/bus/label") is None or f.read("/bus/label").size in (1, t["bus_type"].size)
    s = jg.powerSystem(files["case14"])
    jg.acModel_(s)
    s2 = jg.powerSystem(load_case("case14"))
    jg.acModel_(s2)
    assert np.array_equal(s.model.ac.nodalMatrix.nzval, s2.model.ac.nodalMatrix.nzval)


def test_hdf5_reader_rejects_other_files(jg, tmp_path):
    from juliagrid.jl_amd.hdf5 import H5File
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(ValueError):
        H5File(str(p))


def test_update_bus_type_and_slack_rules(jg):
    """updateBus!(system; label, type) (src/powerSystem/bus.jl:179-206): slack hand-over order and revision counters."""
    import pytest
    s = jg.powerSystem(load_case("case30test"))
    rev = (s.model.revision.type, s.model.revision.slack)
    with pytest.raises(RuntimeError):
        jg.updateBusSystem_(s, label=3, type=3)
    jg.updateBusSystem_(s, label=1, type=2)
    assert s.bus.layout.slack == 0 and s.bus.layout.type[s.bus.label[1] - 1] == 2
    jg.updateBusSystem_(s, label=3, type=3)
    assert s.bus.layout.slack == s.bus.label[3] and s.bus.layout.type[s.bus.label[3] - 1] == 3
    assert s.model.revision.type == rev[0] + 2 and s.model.revision.slack == rev[1] + 2
    jg.updateBusSystem_(s, label=3, type=3)                       # no change, no new revision
    assert s.model.revision.type == rev[0] + 2


def test_add_branch_and_drop_zeros_keep_the_model_equal_to_a_fresh_one(jg, oracle):
    """addBranch! / dropZeros! / updateBranch! on a built AC model (branch.jl:79-167, model.jl:81-110, 342-352): values equal the
    model built from scratch out of the edited tables; the pattern revision moves exactly when the pattern does."""
    from test_reusing_pf_gpu import _tables_of
    s = jg.powerSystem(load_case("case14test"))
    jg.acModel_(s)
    rev = s.model.revision

    def same_as_fresh():
        f = jg.powerSystem(_tables_of(s))
        jg.acModel_(f)
        a, b = s.model.ac, f.model.ac
        assert np.abs(a.nodalMatrix.toscipy().toarray() - b.nodalMatrix.toscipy().toarray()).max() < 1e-14
        assert np.abs(a.nodalMatrixTranspose.toscipy().toarray() - b.nodalMatrix.toscipy().toarray().T).max() < 1e-14
        return b.nodalMatrix.nnz

    p0, nnz0 = rev.acPattern, s.model.ac.nodalMatrix.nnz
    jg.addBranchSystem_(s, from_=2, to=3, resistance=0.02, reactance=0.35)           # parallel to an existing branch: same pattern
    assert rev.acPattern == p0 and s.model.ac.nodalMatrix.nnz == nnz0 == same_as_fresh()
    jg.addBranchSystem_(s, from_=11, to=12, reactance=0.12, turnsRatio=0.95, shiftAngle=-0.17)
    assert rev.acPattern == p0 + 1 and s.model.ac.nodalMatrix.nnz == nnz0 + 2 == same_as_fresh()
    jg.addBranchSystem_(s, from_=16, to=7, resistance=0.01, reactance=0.23, status=0)   # out of service: no entry (branch.jl:143-150)
    assert rev.acPattern == p0 + 1 and s.model.ac.nodalMatrix.nnz == nnz0 + 2
    jg.updateBranchSystem_(s, label=s.branch.number, status=1)                          # ... until it goes into service
    assert rev.acPattern == p0 + 2 and s.model.ac.nodalMatrix.nnz == nnz0 + 4 == same_as_fresh()
    jg.updateBranchSystem_(s, label=s.branch.number, status=0)
    assert s.model.ac.nodalMatrix.nnz == nnz0 + 4                                        # stored zeros stay (model.jl:70-71) ...
    jg.dropZerosSystem_(s)
    assert s.model.ac.nodalMatrix.nnz < nnz0 + 4 and rev.acPattern == p0 + 3             # ... until dropZeros!
    same_as_fresh()
    with pytest.raises(ValueError):
        jg.addBranchSystem_(s, from_=2, to=2, reactance=0.1)
    with pytest.raises(ValueError):
        jg.addBranchSystem_(s, from_=2, to=3)

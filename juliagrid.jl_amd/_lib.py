"""ctypes binding of libjgrid_hip.so (the C ABI declared in include/jgrid.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the compute API
raises.  Nothing here (or anywhere in this package) imports oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JG_LIB") or os.path.join(HERE, "libjgrid_hip.so")     # JG_LIB: another build of the same library (A/B experiments)

I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
I32P = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
I8P = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
F64P = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
U8P = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
VP = C.c_void_p

_lib = None


class JGridError(RuntimeError):
    """ErrorException of the Julia shim: any non-zero return code of the C ABI."""

    def __init__(self, code, msg):
        super().__init__(f"libjgrid_hip error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if not os.environ.get("JG_LIB") and not os.environ.get("JG_ALLOW_STALE"):
        # the library must have been built from the sources that lie next to it (content hash written by build.py beside the .so): a stale
        # binary would pass or fail tests for code that is not in the tree
        from . import build as _build
        if _build.needs_build():
            raise ImportError(f"{LIB_PATH} was not built from the current csrc/ + include/jgrid.h (content hash differs or is missing): rebuild with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`, or set JG_ALLOW_STALE=1 to load it anyway")
    L = C.CDLL(LIB_PATH)
    L.jg_last_error.restype = C.c_char_p
    L.jg_device_count.restype = C.c_int
    sig = {
        "jg_nr_create": [C.POINTER(VP), C.c_int64, I64P, I64P, F64P, F64P, I8P, C.c_int64, C.c_int64, C.c_int64, C.c_int],
        "jg_nr_dims": [VP, I64P],
        "jg_nr_set_injection": [VP, F64P, F64P, C.c_int64],
        "jg_nr_set_voltage": [VP, F64P, F64P, C.c_int64],
        "jg_nr_get_voltage": [VP, F64P, F64P],
        "jg_nr_snapshot_voltage": [VP],
        "jg_nr_restore_voltage": [VP],
        "jg_nr_get_voltage_device": [VP, VP, VP],
        "jg_nr_pack_results_device": [VP, VP],
        "jg_nr_patch_ybus": [VP, C.c_int64, C.c_int64, I64P, F64P],
        "jg_nr_patch_ybus_batch": [VP, C.c_int64, C.c_int64, C.c_int64, I64P, F64P],
        "jg_nr_set_ybus": [VP, F64P, F64P],
        "jg_nr_mismatch": [VP, F64P, F64P],
        "jg_nr_solve": [VP],
        "jg_nr_set_refine": [VP, C.c_int],
        "jg_nr_set_shared": [VP, C.c_int],
        "jg_nr_run": [VP, C.c_int64, C.c_double, I32P, I32P],
        "jg_nr_run_defer": [VP, C.c_int64, C.c_double, C.c_int64, C.POINTER(C.c_int32)],
        "jg_nr_move_lanes": [VP, C.c_int64, VP, I32P, C.POINTER(C.c_int32)],
        "jg_nr_finish": [VP, I32P, I32P],
        "jg_nr_resume": [VP, C.c_int64, C.c_int64, C.c_double, I32P, I32P],
        "jg_nr_pack_rows_device": [VP, VP, C.c_int64, C.c_int64, I32P],
        "jg_nr_get_mismatch": [VP, F64P],
        "jg_nr_get_increment": [VP, F64P],
        "jg_nr_get_jacobian": [VP, F64P],
        "jg_nr_get_maps": [VP, I64P, I64P, I64P, I64P, I64P],
        "jg_nr_get_iteration": [VP, I32P],
        "jg_nr_time_kernel": [VP, C.c_int, C.c_int, C.POINTER(C.c_double)],
        "jg_nr_fast_setup": [VP, F64P, F64P],
        "jg_nr_fast_mismatch": [VP, F64P, F64P],
        "jg_nr_fast_solve": [VP],
        "jg_nr_fast_run": [VP, C.c_int64, C.c_double, I32P, I32P],
        "jg_nr_fast_get_increment": [VP, F64P],
        "jg_nr_set_branches": [VP, C.c_int64, I64P, I64P, I8P, F64P],
        "jg_nr_set_outage_labels": [VP, I64P],
        "jg_nr_branch_quantities": [VP, VP, VP, VP, VP, VP, VP, VP],
        "jg_nr_bus_injection": [VP, F64P],
        "jg_nr_set_screen": [VP, VP],
        "jg_nr_screen": [VP, F64P],
        "jg_nr_screen_device": [VP, VP],
        "jg_nr_screen_rows_device": [VP, VP, C.c_int64, C.c_int64, I32P],
        "jg_gn_create": [C.POINTER(VP), C.c_int64, I64P, I64P, F64P, F64P, C.c_int64, I64P, I64P, F64P, C.c_int64, C.c_int64,
                         I8P, I8P, I64P, C.c_int64, I64P, C.c_int64, C.c_int],
        "jg_gn_dims": [VP, I64P],
        "jg_gn_set_method": [VP, C.c_int],
        "jg_gn_set_measurement": [VP, F64P, F64P, F64P, C.c_int64, C.c_int64],
        "jg_gn_set_voltage": [VP, F64P, F64P, C.c_int64],
        "jg_gn_get_voltage": [VP, F64P, F64P],
        "jg_gn_snapshot_voltage": [VP],
        "jg_gn_restore_voltage": [VP],
        "jg_gn_increment": [VP, F64P],
        "jg_gn_solve": [VP],
        "jg_gn_residual_test": [VP, F64P, I32P],
        "jg_gn_get_normalized_residual": [VP, F64P],
        "jg_gn_evaluate": [VP],
        "jg_gn_set_status": [VP, I8P, I8P],
        "jg_gn_run": [VP, C.c_int64, C.c_double, I32P, I32P],
        "jg_gn_get_maps": [VP, I8P, I64P, I64P],
        "jg_gn_get_jacobian": [VP, F64P],
        "jg_gn_get_residual": [VP, F64P],
        "jg_gn_get_increment": [VP, F64P],
        "jg_gn_get_iteration": [VP, I32P],
        "jg_nr_fast_patch_batch": [VP, C.c_int64, C.c_int64, C.c_int64, I64P, F64P, F64P],
        "jg_gn_set_readings": [VP, C.c_int64, I64P, I8P, F64P, F64P, I8P, F64P, F64P, I8P],
        "jg_gn_draw_noise": [VP, C.c_uint64, C.c_double, C.c_int64],
        "jg_gn_get_measurement": [VP, F64P, F64P, F64P],
        "jg_gn_get_objective": [VP, F64P],
        "jg_gn_pack_results_device": [VP, VP],
        "jg_gn_allgather_results": [VP, VP, VP],
        "jg_gn_time_kernel": [VP, C.c_int, C.c_int, C.POINTER(C.c_double)],
        "jg_plan_create": [C.POINTER(VP), C.c_int64, I32P, I32P, C.c_int64],
        "jg_comm_unique_id": [U8P],
        "jg_comm_create": [C.POINTER(VP), C.c_int64, C.c_int64, U8P, C.c_int],
        "jg_comm_rank": [VP],
        "jg_comm_world": [VP],
        "jg_comm_allgather_device": [VP, VP, VP, C.c_int64],
        "jg_nr_allgather_results": [VP, VP, VP],
        "jg_nr_base_create": [C.POINTER(VP), VP, C.c_int64],
        "jg_nr_base_info": [VP, I64P],
        "jg_nr_base_get": [VP, C.c_int, F64P, C.c_int64],
        "jg_nr_attach_base": [VP, VP],
        "jg_nr_start_from_base": [VP],
        "jg_nr_set_first_iteration": [VP, C.c_int],
        "jg_nr_first_iteration_counts": [VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    L.jg_nr_destroy.argtypes = [VP]
    L.jg_nr_destroy.restype = None
    L.jg_gn_destroy.argtypes = [VP]
    L.jg_gn_destroy.restype = None
    L.jg_nr_base_destroy.argtypes = [VP]
    L.jg_nr_base_destroy.restype = None
    L.jg_plan_cache_clear.argtypes = []
    L.jg_plan_cache_clear.restype = None
    L.jg_comm_destroy.argtypes = [VP]
    L.jg_comm_destroy.restype = None
    L.jg_plan_destroy.argtypes = [VP]
    L.jg_plan_destroy.restype = None
    L.jg_plan_export.argtypes = [VP, C.c_int, VP, C.c_int64]
    L.jg_plan_export.restype = C.c_int64
    L.jg_plan_comp_export.argtypes = [VP, C.c_int64, C.c_int, VP, C.c_int64]
    L.jg_plan_comp_export.restype = C.c_int64
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise JGridError(rc, lib().jg_last_error().decode())


def device_count():
    return int(lib().jg_device_count())


_plan = None


def plan_lib():
    """Library behind `Plan`: libjgrid_hip.so, or -- JG_PLAN_LIB=<path> -- a plan-only build of csrc/jg_symbolic.cpp + jg_plan_api.cpp
    (device-free host code: the CPU suite runs it under AddressSanitizer / UBSan, tools/asan_plan.sh)."""
    global _plan
    if _plan is None:
        path = os.environ.get("JG_PLAN_LIB")
        if not path:
            _plan = lib()
        else:
            _plan = C.CDLL(path)
            _plan.jg_plan_create.argtypes = [C.POINTER(VP), C.c_int64, I32P, I32P, C.c_int64]
            _plan.jg_plan_create.restype = C.c_int
            _plan.jg_plan_destroy.argtypes = [VP]
            _plan.jg_plan_destroy.restype = None
            _plan.jg_plan_export.argtypes = [VP, C.c_int, VP, C.c_int64]
            _plan.jg_plan_export.restype = C.c_int64
            _plan.jg_plan_comp_export.argtypes = [VP, C.c_int64, C.c_int, VP, C.c_int64]
            _plan.jg_plan_comp_export.restype = C.c_int64
    return _plan


class Plan:
    """Device-free symbolic analysis (schedule export for tests)."""

    NAMES = dict(perm=0, e_row=1, e_col=2, e_src=3, t_ptr=4, t_a=5, t_b=6, e_level=7, e_diag=8, diag=9,
                 l_ptr=10, l_ent=11, l_col=12, u_ptr=13, u_ent=14, u_col=15, t_d=16, y_level=17, bwd_level=18, chain_level=19, src_entry=64, bwd_chain=65, pre_pivot=75)

    def __init__(self, n, rowptr, col, policy=0):
        self.h = VP()
        rc = plan_lib().jg_plan_create(C.byref(self.h), int(n), np.ascontiguousarray(rowptr, dtype=np.int32),
                                  np.ascontiguousarray(col, dtype=np.int32), int(policy))
        if rc:
            raise JGridError(rc, "block pattern must be structurally symmetric with a full diagonal")
        self.n = int(n)

    def __del__(self):
        if getattr(self, "h", None):
            plan_lib().jg_plan_destroy(self.h)
            self.h = None

    def get(self, which):
        w = self.NAMES[which] if isinstance(which, str) else int(which)
        m = plan_lib().jg_plan_export(self.h, w, None, 0)
        if m < 0:
            raise KeyError(which)
        out = np.zeros(max(m, 1), dtype=np.int32)
        plan_lib().jg_plan_export(self.h, w, out.ctypes.data, m)
        return out[:m]

    def replay_tables(self, kind):
        """Device replay tables (jg_symbolic.hpp): segments [n,8] = rec_base, nchunks, wpi, rpw, level, last, items, -;
        wave records [m,16]."""
        base = {"fact": 60, "bwd": 62, "fwd": 66, "sel": 68, "pre": 76, "bwdj": 79}[kind]
        return self.get(base).reshape(-1, 8), self.get(base + 1).reshape(-1, 16)

    def comp_tables(self, top_cap):
        """Tables of the shared-factor solve (jg_symbolic.hpp: CompTables): info (n_top, split, forward levels, backward levels), top pivots,
        forward (segments [., 8], records [., 16]), backward (segments, records)."""
        def get(which):
            m = plan_lib().jg_plan_comp_export(self.h, int(top_cap), which, None, 0)
            if m < 0:
                raise KeyError(which)
            out = np.zeros(max(m, 1), dtype=np.int32)
            plan_lib().jg_plan_comp_export(self.h, int(top_cap), which, out.ctypes.data, m)
            return out[:m]
        return get(0), get(1), (get(2).reshape(-1, 8), get(3).reshape(-1, 16)), (get(4).reshape(-1, 8), get(5).reshape(-1, 16))

    def top_tables(self):
        """Multifrontal top (jg_symbolic.hpp): task headers [t,16], task data, launches [l,4] = task_begin, ntasks, class,
        level, grouped, wg_begin, nwg, -; task of each pivot; (top_level, stack doubles per scenario, terms inside tasks, stack doubles per
        interleave class x 3, Jordan plan, blocks of Jordan rows).
        Workgroup map of the grouped launches: get(78)."""
        return self.get(70).reshape(-1, 16), self.get(71), self.get(72).reshape(-1, 8), self.get(73), self.get(74)


COMM_ID_BYTES = 128


class Comm:
    """RCCL communicator of a sharded screen behind the C ABI (include/jgrid.h: jg_comm_*).  Rank 0 draws `Comm.unique_id()`, the host
    ships the 128 bytes to every rank (any channel), every rank calls Comm(rank, world, id, device) -- a collective."""

    @staticmethod
    def unique_id():
        out = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        check(lib().jg_comm_unique_id(out))
        return out

    def __init__(self, rank, world, uid, device=0):
        self.h = VP()
        check(lib().jg_comm_create(C.byref(self.h), int(rank), int(world), np.ascontiguousarray(uid, dtype=np.uint8), int(device)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    def allgather_device(self, send_ptr, recv_ptr, count):
        """count doubles per rank from device pointer send_ptr into recv_ptr [world][count] (send may alias its own block of recv)."""
        check(lib().jg_comm_allgather_device(self.h, VP(int(send_ptr)), VP(int(recv_ptr)), int(count)))

    def close(self):
        if getattr(self, "h", None):
            lib().jg_comm_destroy(self.h)
            self.h = None

    __del__ = close

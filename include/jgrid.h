/*
 * jgrid.h -- C ABI of libjgrid_hip.so: MI355X-native Newton-Raphson AC power flow and Gauss-Newton
 * WLS state estimation inner loops, drop-in behind JuliaGrid's newtonRaphson()/mismatch!()/solve!()
 * and gaussNewton()/increment!()/solve!() (reference paths relative to /root/reference).
 *
 * Conventions
 *  - Every array argument is a HOST pointer owned by the caller and copied during the call; no
 *    pointer outlives the call.  Device memory lives behind the opaque handle.
 *  - Index arrays are the reference's own containers: 1-based int64, CSC (Julia SparseMatrixCSC).
 *  - `batch` independent scenarios of one grid are solved at once.  Per-scenario arrays are
 *    scenario-major on the host ([batch][n], C order); the library keeps them batch-minor in HBM.
 *  - Return codes: 0 ok, 1 bad argument, 2 HIP runtime error, 3 zero / non-finite pivot,
 *    4 stale model.  jg_last_error() gives the text of the last failure on this thread.
 *  - A handle is bound to one device and one HIP stream; handles are independent, a single handle
 *    is not thread-safe.  No global mutable state.
 */
#ifndef JGRID_H
#define JGRID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jg_nr jg_nr;
typedef struct jg_nr_base jg_nr_base;
typedef struct jg_gn jg_gn;

const char* jg_last_error(void);
/* Number of visible HIP devices (<0 on runtime failure). */
int jg_device_count(void);
/* Engines that factorise the same block pattern under the same plan policy on the same device share ONE symbolic analysis and ONE
 * device copy of its replay tables through a process-wide cache (csrc/jg_engine.hpp: SharedPlan; the reference redoes its symbolic
 * factorisation in every newtonRaphson() / gaussNewton() call).  jg_plan_cache_clear drops the cache's own references -- live handles keep
 * their plans -- so that the next jg_*_create pays a full analysis again (benchmarks measure that cost with it).  JG_PLAN_CACHE=0 in the
 * environment switches the cache off. */
void jg_plan_cache_clear(void);

/* ---------------------------------------------------------------------------------------------
 * Newton-Raphson AC power flow
 * ------------------------------------------------------------------------------------------- */

/*
 * newtonRaphson(system)  -- src/powerFlow/acPowerFlow.jl:39-87 after initializeACPowerFlow (:1312-1331).
 * Builds the index maps pq/pvpq and the Jacobian CSC pattern exactly as newtonJacobian (:89-175),
 * runs the symbolic analysis that replaces the first `lu(J)` (src/backend/utility.jl:470-476),
 * and uploads the grid.
 *   n               number of buses
 *   colptr,rowval   Ybus pattern, system.model.ac.nodalMatrix (src/definition/system.jl:213-221)
 *   y_reim          nodalMatrix.nzval as (re,im) pairs, 2*nnz doubles
 *   yt_reim         nodalMatrixTranspose.nzval, same pattern (src/powerSystem/model.jl:75)
 *   type            bus.layout.type AFTER bus-type normalisation (1 PQ, 2 PV, 3 slack)
 *   slack           bus.layout.slack (1-based)
 *   batch           number of scenarios resident on this device (>= 1)
 *   max_patch       Ybus entries a scenario may override (4 per branch outage), >= 0
 *   device          HIP device ordinal
 */
int jg_nr_create(jg_nr** h, int64_t n, const int64_t* colptr, const int64_t* rowval,
                 const double* y_reim, const double* yt_reim, const int8_t* type, int64_t slack,
                 int64_t batch, int64_t max_patch, int device);
void jg_nr_destroy(jg_nr* h);

/* sizes: dims[0]=dimJ, dims[1]=nnz(J), dims[2]=nnz blocks of L+D+U, dims[3]=LU update terms,
 * dims[4]=launches per factorization (forward elimination fused in), dims[5]=launches per backward sweep */
int jg_nr_dims(jg_nr* h, int64_t* dims);

/* bus.supply - bus.demand per scenario (acPowerFlow.jl:676-680). [batch][n] each;
 * batch_stride 0 broadcasts one [n] vector to all scenarios. */
int jg_nr_set_injection(jg_nr* h, const double* p_inj, const double* q_inj, int64_t batch_stride);
/* analysis.voltage.{magnitude,angle} (setInitialPoint!, acPowerFlow.jl:1226-1249, 1281-1295). */
int jg_nr_set_voltage(jg_nr* h, const double* vm, const double* va, int64_t batch_stride);
int jg_nr_get_voltage(jg_nr* h, double* vm, double* va);
/* Device-resident start point: snapshot the current voltages inside HBM / restore them (the
 * setInitialPoint! of a benchmark or Monte-Carlo loop without a PCIe round trip). */
int jg_nr_snapshot_voltage(jg_nr* h);
int jg_nr_restore_voltage(jg_nr* h);
/* Same as jg_nr_get_voltage but into DEVICE buffers of the caller (e.g. for an RCCL gather):
 * vm_dev/va_dev are device pointers, [batch][n] doubles, written on the handle's stream and
 * synchronised before return. */
int jg_nr_get_voltage_device(jg_nr* h, double* vm_dev, double* va_dev);
/* The whole result of a batch as ONE device buffer, [batch][2 n + 2] doubles per scenario: V[n] | theta[n] | iterations |
 * status (the analysis.voltage / method.iteration a caller of powerFlow! reads, src/powerFlow/acPowerFlow.jl:1389-1433) --
 * the record a sharded screen gathers with a single collective. */
int jg_nr_pack_results_device(jg_nr* h, double* dst_dev);

/*
 * Per-scenario Ybus edit on top of the shared base matrix -- what acNodalUpdate!
 * (src/powerSystem/model.jl:81-110) does for updateBranch!(...; status = 0)
 * (src/powerSystem/branch.jl:344-350): k entries, ptr[] = 1-based pointers into nodalMatrix.nzval
 * (entry (row,col)); dy_reim = values ADDED to Y[row,col].  Pattern never changes (stored zeros).
 * Replaces any previous patch of that scenario; k = 0 clears it.
 */
int jg_nr_patch_ybus(jg_nr* h, int64_t scenario, int64_t k, const int64_t* ptr, const double* dy_reim);

/* The same for `count` consecutive scenarios starting at `scenario0` in one call (a contingency screen re-targets a
 * whole batch between solves): ptr [count][k], dy_reim [count][k][2]; ptr 0 = unused slot of that scenario. */
int jg_nr_patch_ybus_batch(jg_nr* h, int64_t scenario0, int64_t count, int64_t k, const int64_t* ptr, const double* dy_reim);

/* Re-upload the SHARED Ybus values after an in-place edit of the system (updateBranch!(analysis; ...),
 * src/powerSystem/branch.jl:453-459 -> acNodalUpdate!, model.jl:81-110).  Same pattern as at create. */
int jg_nr_set_ybus(jg_nr* h, const double* y_reim, const double* yt_reim);

/* mismatch!(analysis) -- acPowerFlow.jl:645-685. max_p/max_q: [batch] infinity norms. */
int jg_nr_mismatch(jg_nr* h, double* max_p, double* max_q);
/* solve!(analysis) -- acPowerFlow.jl:793-911: Jacobian fill, refactorization, solve, state update,
 * iteration += 1, for every scenario. */
int jg_nr_solve(jg_nr* h);
/* Guard of the static-pivot factorisation, opt-in.  mode 1: every Newton step (jg_nr_solve and the steps of jg_nr_run) is followed
 * by ONE step of iterative refinement, rho = f - J d, d += J^-1 rho, before the state moves -- what the reference gets from
 * UMFPACK's solve behind ldiv! (src/backend/utility.jl:576-586; UMFPACK refines by default, KLU does not).  J d is formed from
 * Ybus and the state (the factor has overwritten J); costs one more pass over the rows, a forward-only and a backward sweep
 * (~ +45 % per iteration).  Independent of the mode a pivot block that cancels to rounding level marks its scenario (status 3).
 * mode 0 (default): no refinement -- Newton's iteration corrects a rounding-level error of one step in the next. */
int jg_nr_set_refine(jg_nr* h, int mode);
/* Performance hint, no counterpart in the reference (one analysis at a time).  mode 1: this handle is one of SEVERAL batches in flight on
 * its GPU (a pipeline of contingency batches): the multifrontal top of the factorisation always runs its 4-wave kernel variant, which
 * leaves room on a CU for the workgroups of the other batches (+2-3 % throughput with three 512-scenario batches in flight, -0.6 % for a
 * handle that runs alone).  Results are bitwise the same either way (tests/test_top_variants_gpu.py).  mode 0 (default): by the size of
 * each launch. */
int jg_nr_set_shared(jg_nr* h, int mode);
/* powerFlow!(analysis; iteration, tolerance) -- acPowerFlow.jl:1389-1433, per scenario, with the
 * reference's loop accounting.  iters/status: [batch]; status 0 converged, 1 iteration limit,
 * 3 numeric failure. */
int jg_nr_run(jg_nr* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status);
/* Straggler hand-off between batches of the same grid (no counterpart in the reference, which has no batch: its loop runs one
 * scenario at a time, acPowerFlow.jl:1389-1433).  A batch advances in lockstep until its slowest scenario is done; the last
 * iterations of 512 N-1 scenarios run on a few dozen of them at the latency of a full pass.  A pipeline of batches therefore
 * stops a batch once at most defer_at (<= 64) scenarios are active, moves those into a POOL handle that collects the stragglers of
 * several batches, and finishes them together:
 *   jg_nr_run_defer   jg_nr_run that returns as soon as <= defer_at scenarios are active (n_left of them; 0: the batch is done);
 *                     the handle is PAUSED: lanes packed, results not yet in home order.  Handles of one lane group never pause.
 *   jg_nr_move_lanes  the active scenarios of the paused handle src (state, injections, Ybus patch, iteration count) continue in
 *                     lanes dst_lane0.. of dst; home[i] = the scenario (lane of src) that went to lane dst_lane0 + i, count of them;
 *                     in src they end with status 4 (deferred).  Same grid, same device, dst must not be running.
 *   jg_nr_finish      ends a paused run: lanes home, iters / status [batch] (deferred scenarios: status 4).
 *   jg_nr_resume      runs the scenarios in lanes [0, lanes) of a pool to the end, each with the iteration count it arrived with
 *                     (per-scenario results are bitwise those of an undisturbed batch: lanes never interact, and jg_nr_move_lanes refuses
 *                     a pool that runs another factorisation plan than the batch -- the plan depends on the CLASS of batch a handle was
 *                     created for, decided by the scenario count PADDED to a multiple of 64 lanes: 64 lanes with at most 32 scenarios,
 *                     64 lanes (33-64 scenarios), 128 or 192 lanes (65-192), 256 lanes and more (193 scenarios and more: a batch of
 *                     200 is in the class of 512, not of 192); iters / status [lanes].
 *   jg_nr_pack_rows_device  V | theta | iterations | status of lanes lane0 .. lane0 + count - 1 into rows rows[i] of a result
 *                     record [.][2 n + 2] in device memory (the record jg_nr_pack_results_device writes for the batch they left). */
int jg_nr_run_defer(jg_nr* h, int64_t max_iter, double tol, int64_t defer_at, int32_t* n_left);
int jg_nr_move_lanes(jg_nr* dst, int64_t dst_lane0, jg_nr* src, int32_t* home, int32_t* count);
int jg_nr_finish(jg_nr* h, int32_t* iters, int32_t* status);
int jg_nr_resume(jg_nr* h, int64_t lanes, int64_t max_iter, double tol, int32_t* iters, int32_t* status);
int jg_nr_pack_rows_device(jg_nr* h, double* dst_dev, int64_t lane0, int64_t count, const int32_t* rows);

/*
 * The FIRST iteration of a batch whose scenarios all start from one state, on ONE shared factor (compensation method) -- what replaces, for the
 * user loop of an N-1 screen (src/powerSystem/branch.jl:453-459: updateBranch!(analysis; label, status = 0), setInitialPoint!, powerFlow!), the first
 * lu! + ldiv! of every scenario (src/powerFlow/acPowerFlow.jl:890-897; src/backend/utility.jl:478-484, 576-586).  At the common start the Jacobian of
 * scenario s is J_0 + E M_s F' with M_s the change of the <= 4 blocks of the two buses its Ybus edits touch, so its Newton step is
 *     x_s = J_0^-1 (f_s - E c_s),   c_s = M_s (I + S_s M_s)^-1 F' J_0^-1 f_s,   S_s = the 4 x 4 of J_0^-1 at those buses
 * -- one factorisation per BASE CASE instead of one per scenario; the batch pays a mismatch pass, a correction of <= 4 numbers per scenario and one
 * sweep pair whose factor values are shared (scalar loads).  Iterations >= 2 refactorise as before.  Results agree with the refactorising path to
 * rounding (the step is the same Newton step); a scenario whose 4 x 4 system is singular -- the outage islands a part of the grid -- ends with
 * status 3, like a cancelled pivot of the batched factorisation.
 *   jg_nr_base_create   `single`: a handle with batch = 1 whose CURRENT nodal matrix, injections and voltages are the base case and the common
 *                       start (e.g. its converged power flow).  Factorises its Jacobian once, forms J_0^-1 f_0, the blocks of J_0^-1 on the Ybus
 *                       pattern and the dense inverse of the top of the elimination tree (top_cap: at most this many pivots there; 0 = default
 *                       512, < 0 = none: every level a launch).  The base keeps copies: `single` may be reused or destroyed afterwards.
 *                       rc 3: the base Jacobian is singular.
 *   jg_nr_base_destroy  releases the caller's reference (the memory goes when the last attached handle lets go).
 *   jg_nr_attach_base   the scenarios of h may start from this base (same grid, bus types and device; NULL detaches).  jg_nr_set_ybus detaches.
 *   jg_nr_start_from_base   V, theta of EVERY scenario of h = the base's start (device-side broadcast; replaces jg_nr_restore_voltage in the loop).
 *   jg_nr_run / jg_nr_run_defer then take the shared-factor iteration BY THEMSELVES when (a) the state is untouched since jg_nr_start_from_base,
 *                       (b) every scenario's Ybus edits (jg_nr_patch_ybus*) lie in the rows / columns of at most two buses joined by an edited
 *                       entry -- a branch outage or parameter change, a shunt change --, (c) scenarios with edits keep the base's injections
 *                       (checked on the device once per jg_nr_set_injection; scenarios WITHOUT edits may have any injections: Monte-Carlo
 *                       variations solve J_0 x = f_s directly), (d) no refinement (jg_nr_set_refine); otherwise they refactorise as always.
 *   jg_nr_set_first_iteration   mode 0: always refactorise (the A/B switch of bench.py's value_full_refactor); 1 (default): as above.
 *   jg_nr_first_iteration_counts   how many runs of h started the one way / the other (NULL: not wanted).
 *   jg_nr_base_info     info[8] = pivots in the dense top, forward level above which a pivot belongs to it, forward / backward level launches of a
 *                       sweep pair, the same two without a top (the set-up solver), creation time in microseconds, attached handles.
 *   jg_nr_base_get      test access: which = 0 J_0^-1 on the Ybus pattern [nnz][4] (row-CSR position (i, j) of the stored pattern: block
 *                       (theta_i, V_i) x (P_j, Q_j), row-major), 1 J_0^-1 f_0 [n][2], 2 f_0 [n][2], 3 the dense inverse of the top's Schur
 *                       complement in the fragment order of the kernel that applies it (csrc/jg_comp.hip: k_ctop), 4 the compact factor [entries][4].
 */
int jg_nr_base_create(jg_nr_base** out, jg_nr* single, int64_t top_cap);
void jg_nr_base_destroy(jg_nr_base* b);
int jg_nr_base_info(jg_nr_base* b, int64_t* info);
int jg_nr_base_get(jg_nr_base* b, int which, double* out, int64_t cap);
int jg_nr_attach_base(jg_nr* h, jg_nr_base* b);
int jg_nr_start_from_base(jg_nr* h);
int jg_nr_set_first_iteration(jg_nr* h, int mode);
int jg_nr_first_iteration_counts(jg_nr* h, int64_t* compensated, int64_t* refactorised);

/* analysis.method.{mismatch,increment,jacobian.nzval} in the reference's own ordering
 * (rows/cols pvpq then pq; CSC of newtonJacobian).  [batch][dimJ] / [batch][nnzJ]. */
int jg_nr_get_mismatch(jg_nr* h, double* mism);
int jg_nr_get_increment(jg_nr* h, double* incr);
int jg_nr_get_jacobian(jg_nr* h, double* nzval);
/* analysis.method.{pq,pvpq,pcount} and jacobian.{colptr,rowval} (1-based, bit-exact). */
int jg_nr_get_maps(jg_nr* h, int64_t* pq, int64_t* pvpq, int64_t* pcount, int64_t* jcolptr, int64_t* jrowval);
/* analysis.method.iteration per scenario. */
int jg_nr_get_iteration(jg_nr* h, int32_t* iters);

/*
 * Fast Newton-Raphson (fastNewtonRaphsonBX / XB) on the same handle -- acPowerFlow.jl:215-537 (model), 687-730
 * (mismatch!), 913-983 (solve!), 1389-1433 (powerFlow!).  The two constant matrices are factorised ONCE on the device
 * (as one block matrix diag(B', B'') per bus pair) and every iteration is two forward/backward sweeps.
 * jg_nr_fast_setup: bp, bq [nnz of Ybus] = B'[pvpq r, pvpq c] and B''[pq r, pq c] of the stored Ybus entry (r, c) at
 *   that CSC pointer (fastNewtonJacobian!, :407-447), identity on the diagonal / zero elsewhere for rows and
 *   columns that are not in the reduced matrices (slack; PV buses in B'').  Shared by all scenarios of the batch.
 * jg_nr_fast_mismatch / _solve / _run mirror jg_nr_mismatch / _solve / _run; jg_nr_get_mismatch returns
 *   [active.mismatch | reactive.mismatch], jg_nr_fast_get_increment [active.increment | reactive.increment].
 */
int jg_nr_fast_setup(jg_nr* h, const double* bp, const double* bq);
/* Fast Newton-Raphson under BATCHED outages -- _updateBranch!(::AcPowerFlow{<:FastNewtonRaphson}), src/powerSystem/branch.jl:477 with
 * fastNewtonJacobian! / Pijtheta*, QijV* (acPowerFlow.jl:416-537): the reference edits the entries of B' and B'' a branch touches and refactorises.
 * Here scenario scenario0 + s keeps the shared matrices of jg_nr_fast_setup plus up to k <= 4 edits: ptr [count][k] 1-based pointers into the stored
 * Ybus pattern (0 = unused slot), dbp / dbq [count][k] what is ADDED to B' / B'' at that entry (0 where the entry is not in the reduced matrix).
 * Replaces the edits of those scenarios, keeps the others', then rebuilds and factorises the whole batch ONCE; the iterations are solves only.
 * The Ybus side of the same outage (the mismatches) is jg_nr_patch_ybus_batch.  A scenario whose edited matrix is singular comes back from
 * jg_nr_fast_run with status 3; jg_nr_fast_setup drops all edits. */
int jg_nr_fast_patch_batch(jg_nr* h, int64_t scenario0, int64_t count, int64_t k, const int64_t* ptr, const double* dbp, const double* dbq);
int jg_nr_fast_mismatch(jg_nr* h, double* max_p, double* max_q);
int jg_nr_fast_solve(jg_nr* h);
int jg_nr_fast_run(jg_nr* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status);
int jg_nr_fast_get_increment(jg_nr* h, double* incr);

/*
 * power!(analysis) / current!(analysis) for every scenario of the batch at its CURRENT state --
 * src/postprocessing/acAnalysis.jl:30-169 (power!), 672-704 (current!), formula helpers :838-925.
 * jg_nr_set_branches (once): the branch table the post-processing needs,
 *   from,to [nb] 1-based; status [nb]; param [nb][16] = re,im of nodalFromFrom, nodalFromTo, nodalToFrom, nodalToTo,
 *   admittance (model.jl:54-67), re,im of t_ij = (1/turnsRatio) cis(-shiftAngle) (:846-851), branch conductance,
 *   susceptance, 1/turnsRatio, 0.
 * jg_nr_set_outage_labels: label[batch] = 1-based branch that is out of service in that scenario (0 none): its
 *   quantities are zero there, like an out-of-service branch of the reference (:71-81, :693-701).
 * jg_nr_branch_quantities: any output may be NULL; each [batch][nb][2]:
 *   from_pq, to_pq = PijQij, PjiQji (:898-904); series_pq = PlQl (:906-908); charging_pq = PcQc (:910-919);
 *   from_i, to_i, series_i = (magnitude, angle) of Iij, Iji, Is (:921-931).
 * jg_nr_bus_injection: inj_pq [batch][n][2] = PiQi (:891-896); the injection current, shunt, supply and generator
 *   powers follow from it on the host (juliagrid.jl_amd/powerflow.py:power_, O(n) each).
 */
int jg_nr_set_branches(jg_nr* h, int64_t nb, const int64_t* from, const int64_t* to, const int8_t* status, const double* param);
int jg_nr_set_outage_labels(jg_nr* h, const int64_t* label);
int jg_nr_branch_quantities(jg_nr* h, double* from_pq, double* to_pq, double* series_pq, double* charging_pq,
                            double* from_i, double* to_i, double* series_i);
int jg_nr_bus_injection(jg_nr* h, double* inj_pq);

/*
 * Contingency screen summary on the device (the next widening of the path) -- what a user of the reference reads off power!(analysis)
 * after every powerFlow! of the outage loop (src/powerSystem/branch.jl:453-459 + postprocessing/acAnalysis.jl:30-169), reduced per scenario so that
 * a sharded screen gathers 10 doubles per scenario instead of its 2 n + 2 state record.  For every scenario at its CURRENT state:
 *   rec[b][0] worst loading  max_k max(|S_ij|, |S_ji|) / rating[k]  over the in-service branches with rating[k] > 0 (0 if none), rec[b][1] its branch (1-based, 0: none)
 *   rec[b][2] largest apparent power at a branch end max(|S_ij|, |S_ji|) (pu; PijQij / PjiQji, :898-904), rec[b][3] its branch
 *   rec[b][4] lowest voltage magnitude, rec[b][5] its bus (1-based); rec[b][6] highest, rec[b][7] its bus
 *   rec[b][8] method.iteration, rec[b][9] status (0 converged, 1 iteration limit, 3 numeric failure) of the last jg_nr_run
 * Ties go to the lowest index.  The branch that is out of service in a scenario (jg_nr_set_outage_labels) does not count there.
 * A scenario that ended with status 3 (numeric failure: its state is NaN) delivers NaN in rec[b][0], [2], [4], [6] and index 0 in [1], [3], [5], [7]:
 * rank by rec[b][9] first -- a diverged contingency must never read as the safest one.
 * jg_nr_set_branches with another branch count drops an installed rating (it belongs to the table it was given for): call jg_nr_set_screen again.
 * jg_nr_set_screen: rating [nb] in pu of apparent power (branch.flow.maxFromBus / maxToBus of type 2, src/powerSystem/branch.jl:29-37), 0 = no limit, NULL = none;
 *   needs jg_nr_set_branches.  jg_nr_screen: rec [batch][10] to the host; jg_nr_screen_device: into a device buffer (the operand of jg_comm_allgather_device).
 * jg_nr_screen_rows_device: the summaries of lanes lane0 .. lane0 + count - 1 into rows rows[0 .. count) of a [.][10] device record (a pool handle returns the
 *   stragglers it finished to the record of the batch they came from, like jg_nr_pack_rows_device does for the state record).
 */
int jg_nr_set_screen(jg_nr* h, const double* rating);
int jg_nr_screen(jg_nr* h, double* rec);
int jg_nr_screen_device(jg_nr* h, double* rec_dev);
int jg_nr_screen_rows_device(jg_nr* h, double* rec_dev, int64_t lane0, int64_t count, const int32_t* rows);

/* Measurement hooks (HIP events on the handle's own stream).
 * kernel: 0 fused mismatch+Jacobian assembly, 1 LU refactorization + fused forward elimination (all
 * launches), 2 backward sweep (no state update), 3 power!/current! branch kernel (all outputs), 4 the linear step of a first
 * iteration on the shared base factor (per-scenario correction + sweep pair; needs jg_nr_attach_base), 5 the mismatch-only pass of such
 * a start.  Returns the mean milliseconds of `reps` back-to-back executions. */
int jg_nr_time_kernel(jg_nr* h, int kernel, int reps, double* mean_ms);

/* ---------------------------------------------------------------------------------------------
 * Gauss-Newton WLS state estimation
 * ------------------------------------------------------------------------------------------- */

/*
 * gaussNewton(monitoring) -- src/stateEstimation/acStateEstimation.jl:43-75 over acWLS (:77-259).
 * The caller passes what acWLS derives from the Measurement container row by row (rows ordered
 * voltmeters, ammeters, wattmeters, varmeters, PMUs x2):
 *   code[m]     measurement type code 1..21 (Appendix B of SURVEY.md) BEFORE status masking; codes 22..27 are the
 *               LINEAR rows of pmuStateEstimation (src/stateEstimation/pmuStateEstimation.jl:72-177): state = (Re V, Im V)
 *               per bus (kept in the angle / magnitude arrays), 22/23 bus phasor Re/Im, 24/25 from-end current Re/Im,
 *               26/27 to-end current Re/Im; pass slack = 0 for that model (no reference bus; every variable is estimated),
 *   status[m]   0/1; se.type = status * code (:139, :1139, :1161, :1190-1193, :1222),
 *   index[m]    1-based bus or branch index (se.index),
 *   corr_row[]  1-based FIRST row of every rectangular PMU with a 2x2 precision block (:220-221).
 * The library rebuilds the H pattern exactly as oneIndices!/twoIndices!/fourIndices!/nthIndices!
 * (:1130-1238) + sparse() (:238) do, the block pattern of the gain matrix H'WH, the gather lists
 * and the symbolic analysis replacing the first lu(gain) (src/backend/utility.jl:470-476).
 *   colptr,rowval,y_reim,yt_reim,slack   as for jg_nr_create (system.model.ac)
 *   from,to [nb]                          branch.layout.{from,to}
 *   branch_param [nb][6]                  re(ac.admittance), im(ac.admittance), branch.parameter.conductance,
 *                                         susceptance, turnsRatio, shiftAngle  (equations.jl:147-436)
 */
int jg_gn_create(jg_gn** h, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* y_reim,
                 const double* yt_reim, int64_t nb, const int64_t* from, const int64_t* to, const double* branch_param,
                 int64_t slack, int64_t m, const int8_t* code, const int8_t* status, const int64_t* index,
                 int64_t n_corr, const int64_t* corr_row, int64_t batch, int device);
void jg_gn_destroy(jg_gn* h);
/* dims[0]=m, [1]=nnz(H), [2]=gain blocks, [3]=L+D+U blocks, [4]=LU terms, [5]=factor launches,
 * [6]=backward launches, [7]=H slots (1x2 blocks) */
int jg_gn_dims(jg_gn* h, int64_t* dims);
/* The WlsMethod tag of gaussNewton(monitoring, T) (src/definition/analysis.jl:36-99; increment! methods
 * acStateEstimation.jl:878-971).  method 0 = the Normal tags (LU, KLU, QR, LDLt, LL: gain matrix H'WH, factorised by the
 * block engine).  method 1 = the Orthogonal and PetersWilkinson tags: the least-squares increment of
 * sqrt(W) H d = sqrt(W) r without forming Q -- the triangular factor of the reference's qr(sqrt(W) H) is the Cholesky factor of
 * the gain the engine already holds, and one correction pass through H itself (rho = r - H d; d += G^-1 H'W rho: corrected
 * semi-normal equations) replaces the multiplication by Q'.  Needs a diagonal precision matrix (the reference's
 * sqrtPrecision!, dcStateEstimation.jl:488-492, has the same restriction); returns 1 if the set has correlated PMUs. */
int jg_gn_set_method(jg_gn* h, int method);
/* se.mean [batch][m] and se.precision: diagonal [batch][m] + W[r,r+1] of every correlated pair
 * [batch][n_corr] (acStateEstimation.jl:135-236; equations.jl:576-677).  stride 0 broadcasts. */
int jg_gn_set_measurement(jg_gn* h, const double* mean, const double* wdiag, const double* woff,
                          int64_t batch_stride_m, int64_t batch_stride_corr);
/* Monte-Carlo realisations drawn ON the device.  The reference draws one inside add<Meter>!(...; noise = true) -- mean + variance^(1/2) * randn,
 * src/measurement/utility.jl:70-73 -- and acWLS applies its value rules to the noisy readings (acStateEstimation.jl:135-236; squared currents,
 * rectangular PMUs: equations.jl:576-666).  jg_gn_set_readings (once): the RAW readings per device in acWLS's row order -- row [ndev] 1-based first row,
 * kind [ndev]: 0 one row, value z; 1 one row, squared current (mean z^2, variance 4 z^2 sigma^2); 2 polar PMU (rows magnitude, angle); 3 polar PMU with
 * squared current magnitude; 4 / 5 rectangular PMU without / with its 2x2 precision block (5 exactly on the corr_row rows of jg_gn_create);
 * z1, v1, s1 magnitude (or the single quantity): mean, variance, status; z2, v2, s2 the PMU angle (ignored for kinds 0, 1).
 * jg_gn_draw_noise: lane b becomes realisation first_realisation + b of `seed`: z + scale * sigma * N(0,1) per raw reading from a counter-based generator
 * (csrc/jg_gn.hip: k_gn_noise -- the same realisation gets the same numbers on any rank, batch and lane.  Exactly, for device d (0-based, in row order) and realisation r, in
 * uint64 arithmetic:  c = (seed + 0x9E3779B97F4A7C15 * (2 d)) ^ (r * 0xD1B54A32D192ED03);  u1 = mix64(c), u2 = mix64(c + 0x9E3779B97F4A7C15) with mix64 the splitmix64 finaliser;
 * uniform (0, 1] = ((u >> 11) + 1) 2^-53;  the two normals of the device = sqrt(-2 ln u1) (cos, sin)(2 pi u2) -- the numpy restatement is tests/test_montecarlo_gpu.py:_normals), then the value
 * rules: se.mean and se.precision of every scenario are rewritten in place, nothing crosses PCIe.  scale 0 restores the noise-free set.  Returns 1 when a
 * variance comes out zero or not finite (the reference's errorVariance).  jg_gn_get_measurement: se.mean / diag(se.precision) [batch][m], pair terms [batch][n_corr]. */
int jg_gn_set_readings(jg_gn* h, int64_t ndev, const int64_t* row, const int8_t* kind, const double* z1, const double* v1, const int8_t* s1,
                       const double* z2, const double* v2, const int8_t* s2);
int jg_gn_draw_noise(jg_gn* h, uint64_t seed, double scale, int64_t first_realisation);
int jg_gn_get_measurement(jg_gn* h, double* mean, double* wdiag, double* woff);
int jg_gn_set_voltage(jg_gn* h, const double* vm, const double* va, int64_t batch_stride);
int jg_gn_get_voltage(jg_gn* h, double* vm, double* va);
/* Keep / restore the current state inside HBM (restart of a Monte-Carlo batch from the same start point without a
 * host round trip; the counterpart of jg_nr_snapshot_voltage / jg_nr_restore_voltage). */
int jg_gn_snapshot_voltage(jg_gn* h);
int jg_gn_restore_voltage(jg_gn* h);
/* increment!(analysis) -- acStateEstimation.jl:878-904: residual + Jacobian, gain, factor, solve.
 * max_inc [batch] = maximum(abs, increment). */
int jg_gn_increment(jg_gn* h, double* max_inc);
/* solve!(analysis) -- acStateEstimation.jl:1035-1047 */
int jg_gn_solve(jg_gn* h);
/* stateEstimation!(analysis; iteration, tolerance) -- acStateEstimation.jl:1286-1329, per scenario.
 * status 0 converged, 1 iteration limit, 3 singular gain. */
int jg_gn_run(jg_gn* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status);
/* se.type and jacobian.{colptr,rowval} (m x 2n CSC, 1-based, bit-exact) */
int jg_gn_get_maps(jg_gn* h, int8_t* type, int64_t* hcolptr, int64_t* hrowval);
/* se.jacobian.nzval [batch][nnzH], se.residual [batch][m], se.increment [batch][2n] (theta then V) */
int jg_gn_get_jacobian(jg_gn* h, double* nzval);
int jg_gn_get_residual(jg_gn* h, double* residual);
int jg_gn_get_increment(jg_gn* h, double* increment);
int jg_gn_get_iteration(jg_gn* h, int32_t* iters);
/* se.objective = r' W r at the residual the handle holds (the last increment! / evaluate: after stateEstimation! the residual of every scenario at its
 * final state) -- src/backend/equations.jl:689-698 incl. the cross terms of correlated PMU pairs --, reduced on the device in a fixed order; [batch]. */
int jg_gn_get_objective(jg_gn* h, double* objective);
/*
 * Sharded Monte-Carlo state estimation (SURVEY.md 8e for the Gauss-Newton side): noisy realisations of one measurement set are independent scenarios
 * (the reference draws them in add<Meter>!(...; noise = true), src/measurement/utility.jl:70-73, and estimates them one after the other,
 * acStateEstimation.jl:1286-1329); a rank estimates a contiguous block of them and ONE collective hands every rank the whole result.
 *   jg_gn_pack_results_device  the handle's record after jg_gn_run into device memory of the caller: [batch][2 n + 3] =
 *                              magnitude[n] | angle[n] | method.iteration | status | se.objective  per realisation (the state arrays as jg_gn_get_voltage
 *                              returns them; bitwise the getters' values).  Returns after the stream has drained.
 *   jg_gn_allgather_results    packs into block `rank` of dst_dev [world x batch][2 n + 3] and gathers in place (ncclAllGather of RCCL on the handle's
 *                              stream); every rank calls it with the same batch.
 */
int jg_gn_pack_results_device(jg_gn* h, double* dst_dev);
/* residualTest!(analysis) -- src/stateEstimation/badData.jl:119-311, the numeric part, per scenario: residual, Jacobian,
 * gain and its factor at the CURRENT state, selected inverse of the gain on its factor pattern (replaces
 * takahashiCholeskyLower / selectedInverse, :536-637), c = rowProjection (:289-311), normalised residuals
 * |r_i| / sqrt(|1 / W_ii - c_i|) (0 where r_i == 0 or the row carries no weight in that scenario).
 * max_nres [batch], index [batch] = 1-based row of the largest one (first on ties, 0 if all are zero). */
int jg_gn_residual_test(jg_gn* h, double* max_nres, int32_t* index);
/* update<Meter>!(analysis; label, status) -- src/measurement/powermeter.jl:640-677 and siblings: new in-service mask,
 * se.type = status * code; the Jacobian pattern (and every table derived from it) stays.  code (optional, NULL = keep):
 * new type codes; only 2 <-> 4 and 3 <-> 5 may change (updateAmmeter!(...; square), ammeter.jl:367-420). */
int jg_gn_set_status(jg_gn* h, const int8_t* status, const int8_t* code);
/* residual and Jacobian at the CURRENT state, nothing else (se.residual for chiTest after the last solve!) */
int jg_gn_evaluate(jg_gn* h);
/* all normalised residuals of the last jg_gn_residual_test [batch][m] */
int jg_gn_get_normalized_residual(jg_gn* h, double* nres);
/* kernel: 0 measurement rows (H + residual), 1 gain + rhs gather, 2 factor (+ fused forward), 3 backward,
 * 4 selected inverse (needs a factor: call after an increment) */
int jg_gn_time_kernel(jg_gn* h, int kernel, int reps, double* mean_ms);

/* ---------------------------------------------------------------------------------------------
 * Sharded contingency screen: the final gather.  Scenarios are independent, a rank (one process per GPU) solves a
 * contiguous block of them, and ONE collective -- ncclAllGather of RCCL over xGMI -- hands every rank the result record of the whole
 * screen in scenario order.  The reference has no counterpart: its user-level loop runs the scenarios one after the other in one
 * process (src/powerSystem/branch.jl:453-459: updateBranch!(...; status = 0), powerFlow!, updateBranch!(...; status = 1)).
 *   jg_comm_unique_id  rank 0 draws the 128-byte id of a communicator; the HOST ships it to the other ranks (MPI, a file, a socket)
 *   jg_comm_create     collective over the `world` ranks: rank's communicator on HIP device `device`
 *   jg_nr_allgather_results  packs the handle's record ([batch][2 n + 2]: V | theta | iterations | status, jg_nr_pack_results_device)
 *                      into block `rank` of dst_dev [world x batch][2 n + 2] (device memory of the caller) and gathers in place on
 *                      the handle's stream; every rank must call it with the same batch.  Returns after the stream has drained.
 *   jg_comm_allgather_device  the same collective for a record that is already packed (a ContingencyPipeline fills its records
 *                      itself): count doubles per rank, recv_dev [world][count]; send_dev may be recv_dev + rank * count.  The gather runs on the
 *                      communicator's own stream: whatever produced send_dev must have been synchronised before the call; it returns when done.
 * librccl is bound at run time on the first call (csrc/jg_comm.cpp); return code 2 with jg_last_error() when it is missing.
 * ------------------------------------------------------------------------------------------- */
#define JG_COMM_ID_BYTES 128
typedef struct jg_comm jg_comm;
int jg_comm_unique_id(uint8_t* id);
int jg_comm_create(jg_comm** c, int64_t rank, int64_t world, const uint8_t* id, int device);
void jg_comm_destroy(jg_comm* c);
int jg_comm_rank(const jg_comm* c);
int jg_comm_world(const jg_comm* c);
int jg_comm_allgather_device(jg_comm* c, const double* send_dev, double* recv_dev, int64_t count);
int jg_nr_allgather_results(jg_nr* h, jg_comm* c, double* dst_dev);
int jg_gn_allgather_results(jg_gn* h, jg_comm* c, double* dst_dev);

/* ---------------------------------------------------------------------------------------------
 * Symbolic analysis only (no device needed): the static schedule that replaces the symbolic half
 * of `lu`/`klu` (src/backend/utility.jl:470-476, 486-492).  Used by the CPU test-suite to replay
 * and race-check the schedule.  pattern: 0-based int32 block CSR, structurally symmetric, full
 * diagonal.  policy: bit 0 in-place factor storage, bit 1 symmetric values (LDL'), bit 2 the producer finishes level 0 (the
 * leaf pivots: factorised diagonal blocks + rhs rows; csrc/jg_symbolic.hpp), bits 4-7 / 8-15 / 16-23 / 24-30 where the
 * multifrontal top starts and how large its fronts get (0 = defaults), bits 32-39 / 40-47 / 48 the grouped tasks below the
 * top (csrc/jg_symbolic.hpp: "mid"), bit 49 Jordan rows for the pivots of the top tasks + the backward tables over them (what
 * jg_nr_create / jg_gn_create ask for; csrc/jg_symbolic.hpp).  jg_plan_export(which): see csrc/jg_plan_api.cpp;
 * out == NULL returns the length.
 * ------------------------------------------------------------------------------------------- */
typedef struct jg_plan jg_plan;
int jg_plan_create(jg_plan** p, int64_t n, const int32_t* rowptr, const int32_t* col, int64_t policy);
void jg_plan_destroy(jg_plan* p);
int64_t jg_plan_export(jg_plan* p, int which, int32_t* out, int64_t cap);
/* The tables of the shared-factor solve behind jg_nr_base_create (csrc/jg_symbolic.hpp: CompTables) for a dense top of at most top_cap pivots (< 0: none):
 * which = 0 {top pivots, split level, forward levels, backward levels}, 1 the top's pivots, 2 / 3 forward segments (x 8) / records (x 16), 4 / 5 backward. */
int64_t jg_plan_comp_export(jg_plan* p, int64_t top_cap, int which, int32_t* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* JGRID_H */

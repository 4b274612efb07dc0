// jg_gn.hip -- Gauss-Newton WLS state estimation on MI355X: C ABI (include/jgrid.h), measurement
// kernel, gain assembly by gather lists, per-scenario loop control.
//
// Reference behaviour restated (paths relative to /root/reference):
//   acWLS (H pattern, index builders)   src/stateEstimation/acStateEstimation.jl:77-259, 1130-1238
//   normalEquation!                     src/stateEstimation/acStateEstimation.jl:261-583
//   measurement functions               src/backend/equations.jl:20-60, 147-573
//   increment!{Normal}                  src/stateEstimation/acStateEstimation.jl:878-904
//   solve! / stateEstimation!           src/stateEstimation/acStateEstimation.jl:1035-1047, 1286-1329
//
// Design (not a translation).  The reference fills H column by column with a binary-search
// setindex! per entry and recomputes branch coefficients per entry, then forms H'WH with two
// allocating sparse products.  Here one wave evaluates one measurement ROW for 64 scenarios: a
// single sincos (or sincos pair) per row yields the value, the residual and every partial of that
// row, written as 1x2 (d/dtheta, d/dV) blocks per touched bus.  The gain matrix is an n x n
// matrix of 2x2 blocks ((theta_i, V_i) per bus) assembled by precomputed GATHER lists (every block
// knows which (weight, slot, slot) products land in it): deterministic, no atomics, coalesced
// 512-byte segments; its factorisation and the solve reuse the block LU engine of the NR path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/jgrid.h"
#include "jg_engine.hpp"

namespace {

int failg(int code, const std::string& msg) { jg::set_last_error(msg); return code; }

#define GN_HIP(expr)                                                                    \
    do {                                                                                \
        hipError_t err__ = (expr);                                                      \
        if (err__ != hipSuccess) return failg(2, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    } while (0)

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct RowDesc { int type; int idx; int slot0; int nslots; };      // idx: bus or branch (0-based)
struct BranchP { double g, b, gs, bs, tinv, shift; int from, to; double pad; };   // 64 bytes

struct RowArgs {
    const RowDesc* rows; const int* slot_bus; const BranchP* br;
    const int* rowptr; const double* G; const double* Bv; const int* ydiag;   // Ybus CSR (Y[i,j]) + diagonal pointers
    const double* vm; const double* va; const double* mean;
    double* Hs; double* res;
    int m; int ld;
    const int* items; int n_items;              // work items of k_gn_rows (below): {row | kind << 28, row, row, row}
};

constexpr int GN_ROWS = 16;     // measurement rows per workgroup (4 waves x 4)
#ifndef JG_GN_ITEMS
#define JG_GN_ITEMS 32
#endif
constexpr int GN_ITEMS = JG_GN_ITEMS;   // work items of k_gn_rows per workgroup (4 waves; 4 / 8 / 16 / 32: 0.83 / 0.77 / 0.70-0.74 / 0.69 ms on config 4)

// active / reactive power flow at one end of a branch (types 7, 8, 10, 11) from s, c = sin, cos(theta_i - theta_j - shift): the four rows of a branch share
// them (equations.jl:147-277)
__device__ __forceinline__ void flow_row(int ty, const BranchP& p, double Vi, double Vj, double s, double c,
                                         double& h, double& ti, double& vi, double& tj, double& vj) {
#pragma clang fp contract(off)
    const double g = p.g, b = p.b, gs = p.gs, bs = p.bs, tv = p.tinv;
    const double B = tv * g, C = tv * b;
    if (ty == 7) { const double A = tv * tv * (g + gs);
        h = A * Vi * Vi - (B * c + C * s) * Vi * Vj; ti = (B * s - C * c) * Vi * Vj; vi = 2 * A * Vi - (B * c + C * s) * Vj; tj = -ti; vj = -(B * c + C * s) * Vi;
    } else if (ty == 8) { const double A = g + gs;
        h = A * Vj * Vj - (B * c - C * s) * Vi * Vj; ti = (B * s + C * c) * Vi * Vj; vi = (-B * c + C * s) * Vj; tj = -ti; vj = 2 * A * Vj - (B * c - C * s) * Vi;
    } else if (ty == 10) { const double A = tv * tv * (b + bs);
        h = -A * Vi * Vi - (B * s - C * c) * Vi * Vj; ti = -(B * c + C * s) * Vi * Vj; vi = -2 * A * Vi - (B * s - C * c) * Vj; tj = -ti; vj = -(B * s - C * c) * Vi;
    } else { const double A = b + bs;
        h = -A * Vj * Vj + (B * s + C * c) * Vi * Vj; ti = (B * c - C * s) * Vi * Vj; vi = (B * s + C * c) * Vj; tj = -ti; vj = -2 * A * Vj + (B * s + C * c) * Vi;
    }
}

// value and partials of one branch measurement; i = from, j = to (equations.jl:147-547)
__device__ __forceinline__ void branch_row(int ty, const BranchP& p, double Vi, double Vj, double thi, double thj,
                                           double& h, double& ti, double& vi, double& tj, double& vj) {
    // No contraction into fused multiply-adds here.  The reference's formulas for a current magnitude form I^2 = A Vi^2 + B Vj^2 - 2 Vi Vj (C cos -+ D sin)
    // from terms of size (|y| V)^2 -- 1e6 .. 1e9 -- that cancel to ~1e-2 (equations.jl:279-458); Julia does not contract, and a fused term differs from
    // the reference's by one rounding that the cancellation amplifies to 6e-7 of the row's scale in H.  Measured with the library built both ways
    // (tools/se_contract_probe.sh, profiles/r04_se_contract_probe.txt): rows of type 2 / 3 |dH| 1.4e-3 -> 7e-13, types 14 / 15 6e-7 -> 7e-8 at |H| 1e4.
#pragma clang fp contract(off)
    const double g = p.g, b = p.b, gs = p.gs, bs = p.bs, tv = p.tinv;
    if (ty >= 18) {                                   // rectangular current phasors (types 18-21)
        double si, ci, sj, cj, A, B;
        const double C = tv * g, D = tv * b;
        if (ty == 18 || ty == 20) {                   // psi_ij coefficients, ViVjthetaithetajState
            A = tv * tv * (g + gs); B = tv * tv * (b + bs);
            sincos(thi, &si, &ci); sincos(thj + p.shift, &sj, &cj);
            if (ty == 18) { h = (A * ci - B * si) * Vi - (C * cj - D * sj) * Vj; ti = -(A * si + B * ci) * Vi; vi = A * ci - B * si; tj = (C * sj + D * cj) * Vj; vj = -C * cj + D * sj; }
            else { h = (A * si + B * ci) * Vi - (C * sj + D * cj) * Vj; ti = (A * ci - B * si) * Vi; vi = A * si + B * ci; tj = (-C * cj + D * sj) * Vj; vj = -C * sj - D * cj; }
        } else {                                      // psi_ji coefficients, VjVithetajthetaiState
            A = g + gs; B = b + bs;
            sincos(thi - p.shift, &si, &ci); sincos(thj, &sj, &cj);
            if (ty == 19) { h = (A * cj - B * sj) * Vj - (C * ci - D * si) * Vi; ti = (C * si + D * ci) * Vi; vi = -C * ci + D * si; tj = -(A * sj + B * cj) * Vj; vj = A * cj - B * sj; }
            else { h = (A * sj + B * cj) * Vj - (C * si + D * ci) * Vi; ti = (-C * ci + D * si) * Vi; vi = -C * si - D * ci; tj = (A * cj - B * sj) * Vj; vj = A * sj + B * cj; }
        }
        return;
    }
    double s, c;
    sincos(thi - thj - p.shift, &s, &c);              // ViVjthetaijState
    if (ty == 7 || ty == 8 || ty == 10 || ty == 11) { flow_row(ty, p, Vi, Vj, s, c, h, ti, vi, tj, vj); return; }
    // current magnitude / squared magnitude / angle: I_ij and I_ji coefficient sets
    const double t2 = tv * tv;
    double A, B, C, D, sg;                           // sg: sign of the D term inside (C cos -/+ D sin)
    const bool fromEnd = (ty == 2 || ty == 4 || ty == 14);
    if (fromEnd) { A = t2 * t2 * ((g + gs) * (g + gs) + (b + bs) * (b + bs)); B = t2 * (g * g + b * b); C = t2 * tv * (g * (g + gs) + b * (b + bs)); D = t2 * tv * (g * bs - b * gs); sg = -1.0; }
    else { A = t2 * (g * g + b * b); B = (g + gs) * (g + gs) + (b + bs) * (b + bs); C = tv * (g * (g + gs) + b * (b + bs)); D = tv * (g * bs - gs * b); sg = 1.0; }
    const double e = C * c + sg * D * s;              // (C cos - D sin) for ij, (C cos + D sin) for ji
    const double f = C * s - sg * D * c;              // (C sin + D cos) for ij, (C sin - D cos) for ji
    if (ty == 2 || ty == 3) {
        const double Iinv = 1.0 / sqrt(A * Vi * Vi + B * Vj * Vj - 2 * Vi * Vj * e);
        h = 1.0 / Iinv; ti = Iinv * f * Vi * Vj; vi = Iinv * (A * Vi - e * Vj); tj = -ti; vj = Iinv * (B * Vj - e * Vi);
    } else if (ty == 4 || ty == 5) {
        h = A * Vi * Vi + B * Vj * Vj - 2 * Vi * Vj * e; ti = 2 * f * Vi * Vj; vi = 2 * (A * Vi - e * Vj); tj = -ti; vj = 2 * (B * Vj - e * Vi);
    } else {                                          // 14 psi_ij, 15 psi_ji: angle of the phasor, partials from the magnitude model
        double si, ci, sj, cj, re, im;
        const double Cp = tv * g, Dp = tv * b;
        if (ty == 14) {
            const double Ap = t2 * (g + gs), Bp = t2 * (b + bs);
            sincos(thi, &si, &ci); sincos(thj + p.shift, &sj, &cj);
            re = (Ap * ci - Bp * si) * Vi - (Cp * cj - Dp * sj) * Vj; im = (Ap * si + Bp * ci) * Vi - (Cp * sj + Dp * cj) * Vj;
        } else {
            const double Ap = g + gs, Bp = b + bs;
            sincos(thi - p.shift, &si, &ci); sincos(thj, &sj, &cj);
            re = (Ap * cj - Bp * sj) * Vj - (Cp * ci - Dp * si) * Vi; im = (Ap * sj + Bp * cj) * Vj - (Cp * si + Dp * ci) * Vi;
        }
        const double Iinv2 = 1.0 / (re * re + im * im);
        h = atan2(im, re);
        ti = Iinv2 * (A * Vi * Vi - e * Vi * Vj); vi = -Iinv2 * f * Vj; tj = Iinv2 * (B * Vj * Vj - e * Vi * Vj); vj = Iinv2 * f * Vi;
    }
}

// One wave = one WORK ITEM x 64 scenarios.  An item is a measurement row, or rows that share their operands (built once in jg_gn_create):
//   kind 1  the power-flow rows of ONE branch (wattmeter / varmeter at either end: types 7, 8, 10, 11, up to four rows): V and theta of the two buses are
//           read once and sin, cos(theta_ij) computed once for all of them -- the reference evaluates every row on its own (acStateEstimation.jl:261-583);
//   kind 2  active + reactive injection at ONE bus (types 6, 9): one sweep over the bus's Ybus row (V, theta, sin, cos per neighbour) for both rows.
// Every row keeps its own formulas and the order of its sums, so H and the residual are what the single rows give.  A row that is out of service
// (type 0) inside a group leaves zeros, as on its own.  BASELINE config 4: 82 678 of the 96 723 rows are flow or injection rows -> 25 290 groups.
__global__ __launch_bounds__(256, 4) void k_gn_rows(RowArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * GN_ITEMS;
    const int r1 = min(r0 + GN_ITEMS, a.n_items);
    // the row table, the slot -> bus map, the branch parameters and Ybus are wave-uniform and immutable within a launch: they come
    // through the scalar cache (constant address space), not as 64 identical vector loads followed by readfirstlane
    typedef const int __attribute__((address_space(4)))* CInt;
    typedef const double __attribute__((address_space(4)))* CDbl;
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef const i4 __attribute__((address_space(4)))* CInt4;
    typedef int i16 __attribute__((ext_vector_type(16)));
    typedef const i16 __attribute__((address_space(4)))* CBranch;       // a BranchP is 64 bytes: one s_load_dwordx16
    auto branch = [&](int k) { const i16 raw = ((CBranch)a.br)[k]; BranchP q; __builtin_memcpy(&q, &raw, sizeof(q)); return q; };
    CInt slot_bus = (CInt)a.slot_bus, rowptr = (CInt)a.rowptr, ydiag = (CInt)a.ydiag;
    CDbl Gs = (CDbl)a.G, Bs = (CDbl)a.Bv;
    for (int pos = r0 + wave; pos < r1; pos += blockDim.y) {
        const i4 item = ((CInt4)a.items)[pos];
        const int kind = item[0] >> 28;
        if (kind == 1) {                               // the flow rows of one branch
            const i4 first = ((CInt4)a.rows)[item[0] & 0x0fffffff];
            const BranchP p = branch(first[1]);
            const int i = p.from, j = p.to;
            const double Vi = a.vm[(size_t)i * ld + b], thi = a.va[(size_t)i * ld + b];
            const double Vj = a.vm[(size_t)j * ld + b], thj = a.va[(size_t)j * ld + b];
            double sn, cs;
            sincos(thi - thj - p.shift, &sn, &cs);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rk = k == 0 ? (item[0] & 0x0fffffff) : item[k];
                if (rk < 0) break;
                const i4 rd = k == 0 ? first : ((CInt4)a.rows)[rk];
                double h = 0.0, ti = 0.0, vi = 0.0, tj = 0.0, vj = 0.0, z = 0.0;
                if (rd[0] != 0) { z = a.mean[(size_t)rk * ld + b]; flow_row(rd[0], p, Vi, Vj, sn, cs, h, ti, vi, tj, vj); }
                jg::store_vec(a.Hs, (size_t)rd[2], b, ld, ti, vi);
                jg::store_vec(a.Hs, (size_t)rd[2] + 1, b, ld, tj, vj);
                a.res[(size_t)rk * ld + b] = z - h;
            }
            continue;
        }
        if (kind == 2) {                               // active + reactive injection at one bus: slots follow the Ybus row of the bus
            const int ra = item[0] & 0x0fffffff, rb = item[1];
            const i4 da = ((CInt4)a.rows)[ra], db = ((CInt4)a.rows)[rb];
            const bool ona = da[0] != 0, onb = db[0] != 0;
            const int i = da[1], ns = da[3];
            const double Vi = a.vm[(size_t)i * ld + b], thi = a.va[(size_t)i * ld + b];
            const int p0 = rowptr[i];
            const int pd = ydiag[i];
            double s1 = 0.0, s2 = 0.0;
            for (int s = 0; s < ns; ++s) {
                const int pp = p0 + s;
                const int j = slot_bus[da[2] + s];
                const double g = Gs[pp], bb = Bs[pp];
                const double Vj = a.vm[(size_t)j * ld + b], thj = a.va[(size_t)j * ld + b];
                double sn, cs;
                sincos(thi - thj, &sn, &cs);
                const double ac = g * cs + bb * sn, ad = g * sn - bb * cs;
                s1 += Vj * ac; s2 += Vj * ad;
                if (pp != pd) {
                    jg::store_vec(a.Hs, (size_t)(da[2] + s), b, ld, ona ? Vi * Vj * ad : 0.0, ona ? Vi * ac : 0.0);
                    jg::store_vec(a.Hs, (size_t)(db[2] + s), b, ld, onb ? -(Vi * Vj) * ac : 0.0, onb ? Vi * ad : 0.0);
                }
            }
            const double gii = Gs[pd], bii = Bs[pd];
            const int sd = pd - p0;
            jg::store_vec(a.Hs, (size_t)(da[2] + sd), b, ld, ona ? -Vi * s2 - bii * (Vi * Vi) : 0.0, ona ? s1 + gii * Vi : 0.0);
            jg::store_vec(a.Hs, (size_t)(db[2] + sd), b, ld, onb ? Vi * s1 - gii * (Vi * Vi) : 0.0, onb ? s2 - bii * Vi : 0.0);
            a.res[(size_t)ra * ld + b] = ona ? a.mean[(size_t)ra * ld + b] - Vi * s1 : 0.0;
            a.res[(size_t)rb * ld + b] = onb ? a.mean[(size_t)rb * ld + b] - Vi * s2 : 0.0;
            continue;
        }
        const int r = item[0];
        const i4 rd = ((CInt4)a.rows)[r];
        const int ty = rd[0], idx = rd[1], s0 = rd[2], ns = rd[3];
        // a slot = the 1x2 block (d/dtheta, d/dV) of one (row, bus) pair, interleaved per scenario: one 16-byte store
        auto put = [&](int s, double dth, double dvm) { jg::store_vec(a.Hs, (size_t)(s0 + s), b, ld, dth, dvm); };
        if (ty == 0) {                                 // masked: row kept, H = 0, residual 0 (T7)
            for (int s = 0; s < ns; ++s) put(s, 0.0, 0.0);
            a.res[(size_t)r * ld + b] = 0.0;
            continue;
        }
        const double z = a.mean[(size_t)r * ld + b];
        if (ty >= 22) {                                // LINEAR rows of the PMU-only model: the state is (Re V, Im V) per bus, kept in (va, vm)
            if (ty <= 23) {                            // bus phasor (pmuStateEstimation.jl:128-139)
                const double x = ty == 22 ? a.va[(size_t)idx * ld + b] : a.vm[(size_t)idx * ld + b];
                put(0, ty == 22 ? 1.0 : 0.0, ty == 22 ? 0.0 : 1.0);
                a.res[(size_t)r * ld + b] = z - x;
            } else {                                   // branch current phasor (:140-166, backend/expressions.jl:291-349)
                const BranchP p = branch(idx);
                const int i = p.from, j = p.to;
                double sn, cs, A, B, Cc, D;
                sincos(p.shift, &sn, &cs);
                if (ty <= 25) { A = p.tinv * p.tinv * (p.g + p.gs); B = -p.tinv * p.tinv * (p.b + p.bs); Cc = -p.tinv * (p.g * cs - p.b * sn); D = p.tinv * (p.b * cs + p.g * sn); }
                else { A = -p.tinv * (p.g * cs + p.b * sn); B = p.tinv * (p.b * cs - p.g * sn); Cc = p.g + p.gs; D = -p.b - p.bs; }
                const double ri = a.va[(size_t)i * ld + b], xi = a.vm[(size_t)i * ld + b];
                const double rj = a.va[(size_t)j * ld + b], xj = a.vm[(size_t)j * ld + b];
                if (ty == 24 || ty == 26) { put(0, A, B); put(1, Cc, D); a.res[(size_t)r * ld + b] = z - (A * ri + B * xi + Cc * rj + D * xj); }
                else { put(0, -B, A); put(1, -D, Cc); a.res[(size_t)r * ld + b] = z - (-B * ri + A * xi - D * rj + Cc * xj); }
            }
            continue;
        }
        if (ty == 1 || ty == 12 || ty == 13 || ty == 16 || ty == 17) {
            const double V = a.vm[(size_t)idx * ld + b], th = a.va[(size_t)idx * ld + b];
            double h, dt, dv;
            if (ty == 13) { h = th; dt = 1.0; dv = 0.0; }
            else if (ty == 16) { double s, c; sincos(th, &s, &c); h = V * c; dt = -V * s; dv = c; }
            else if (ty == 17) { double s, c; sincos(th, &s, &c); h = V * s; dt = V * c; dv = s; }
            else { h = V; dt = 0.0; dv = 1.0; }
            put(0, dt, dv);
            a.res[(size_t)r * ld + b] = z - h;
        } else if (ty == 6 || ty == 9) {               // injections: slots follow the Ybus row of the bus
            const int i = idx;
            const double Vi = a.vm[(size_t)i * ld + b], thi = a.va[(size_t)i * ld + b];
            const int p0 = rowptr[i];
            const int pd = ydiag[i];
            double s1 = 0.0, s2 = 0.0;
            for (int s = 0; s < ns; ++s) {
                const int p = p0 + s;
                const int j = slot_bus[s0 + s];
                const double g = Gs[p], bb = Bs[p];
                const double Vj = a.vm[(size_t)j * ld + b], thj = a.va[(size_t)j * ld + b];
                double sn, cs;
                sincos(thi - thj, &sn, &cs);
                const double ac = g * cs + bb * sn, ad = g * sn - bb * cs;
                s1 += Vj * ac; s2 += Vj * ad;
                if (p != pd) {
                    if (ty == 6) put(s, Vi * Vj * ad, Vi * ac);
                    else put(s, -(Vi * Vj) * ac, Vi * ad);
                }
            }
            const double gii = Gs[pd], bii = Bs[pd];
            const int sd = pd - p0;
            if (ty == 6) { put(sd, -Vi * s2 - bii * (Vi * Vi), s1 + gii * Vi); a.res[(size_t)r * ld + b] = z - Vi * s1; }
            else { put(sd, Vi * s1 - gii * (Vi * Vi), s2 - bii * Vi); a.res[(size_t)r * ld + b] = z - Vi * s2; }
        } else {                                       // branch rows: slots = [from, to]
            const BranchP p = branch(idx);
            const int i = p.from, j = p.to;
            const double Vi = a.vm[(size_t)i * ld + b], thi = a.va[(size_t)i * ld + b];
            const double Vj = a.vm[(size_t)j * ld + b], thj = a.va[(size_t)j * ld + b];
            double h, ti, vi, tj, vj;
            branch_row(ty, p, Vi, Vj, thi, thj, h, ti, vi, tj, vj);
            put(0, ti, vi); put(1, tj, vj);
            a.res[(size_t)r * ld + b] = z - h;
        }
    }
}

// Gather-list assembly of the gain blocks and of H'W res as WAVE RECORDS (the idiom of the LU engine, jg_symbolic.hpp): an item
// (one gain block, or one rhs row) is a run of 64-byte records {head, dst, n, -, (weight, slot, slot | row) x 4}; a wave owns a
// contiguous run of records (whole items, bus rows in pivot order) and fetches each with ONE scalar load, the next one while
// the 12 operand loads of the current one are in flight.  (The first build chased item -> three index lists -> values and
// finished the last < 4 contributions one by one: two dependent round trips per contribution for the many 2-term blocks of
// two-hop neighbours.)  Sums run in list order: bitwise the same result as before.
//   head: bit 0 kind (0 gain block, 1 rhs row), bit 4 row is the slack bus, bit 5 column is the slack bus, bit 8 last record
constexpr int GAIN_T = 4;
struct GainRec { int w[16]; };
typedef int GRecS __attribute__((ext_vector_type(16)));
typedef const GRecS __attribute__((address_space(4)))* GRecPtr;
struct GainArgs {
    const GainRec* rec; const int* wave_ptr;      // wave v owns records [wave_ptr[v], wave_ptr[v + 1])
    const double* Hs; const double* res; const double* w;
    double* Gv; double* rhs;
    int n_waves; int ld;
};

__global__ __launch_bounds__(256) void k_gn_gain(GainArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    int grp, bx;
    if (!jg::map_block(jg::GroupSel{}, a.ld, (a.n_waves + 3) / 4, grp, bx)) return;    // scenario group = fast grid index: one group, one XCD
    const int wv = bx * 4 + wave;
    if (wv >= a.n_waves) return;
    const size_t b = (size_t)grp * 64 + lane;
    typedef const int __attribute__((address_space(4)))* CInt;
    CInt wp = (CInt)a.wave_ptr;
    int r = wp[wv];
    const int r1 = wp[wv + 1];
    if (r >= r1) return;
    GRecS cur = ((GRecPtr)a.rec)[r];
    double g00 = 0.0, g01 = 0.0, g10 = 0.0, g11 = 0.0;
    for (; r < r1; ++r) {
        GRecS nxt = cur;
        if (r + 1 < r1) nxt = ((GRecPtr)a.rec)[r + 1];
        const int head = cur[0], n = cur[2];
        const bool is_rhs = head & 1;
        // Hand-placed requests, one wait (jg_engine.hpp: gload16; round 4).  The first build chose between three kinds of load per term behind
        // run-time branches -- residual row, second slot, or none where both slots are the same -- and hipcc waited for every single load before
        // it requested the next one (rocprof ISA: `global_load ...; s_waitcnt vmcnt(0)` per operand): a record's operands arrived one by one.
        static_assert(GAIN_T == 4, "operand list of the wait");
        double wt[GAIN_T], yr[GAIN_T];
        jg::d2v xa[GAIN_T], xb[GAIN_T];
        const unsigned off8 = (unsigned)b * 8u, off16 = (unsigned)b * 16u;
#pragma unroll
        for (int t = 0; t < GAIN_T; ++t) {
            // ONE statement per term, its three kinds of right-hand operand behind SCALAR branches inside it (mode 0: the same slot on both sides -- half of
            // all block terms --, 1: a residual, 2: a second slot): every destination is defined by exactly one statement on every path.  With the
            // choice written as C++ branches around separate requests the compiler merged the alternatives into one register tuple and COPIED a
            // register whose load was still in flight (tools/check_asm_loads.py found it; the pmu case of tests/test_se_gpu.py failed on it).
            const int mode = t < n ? (is_rhs ? 1 : (cur[6 + 3 * t] != cur[5 + 3 * t] ? 2 : 0)) : 3;
            const char* pw = (const char*)a.w + (size_t)cur[4 + 3 * t] * ld * 8;
            const char* px = (const char*)a.Hs + (size_t)cur[5 + 3 * t] * ld * 16;
            const char* py = (mode == 1 ? (const char*)a.res + (size_t)cur[6 + 3 * t] * ld * 8 : (const char*)a.Hs + (size_t)cur[6 + 3 * t] * ld * 16);
            asm volatile("s_cmp_eq_u32 %9, 3\n\t"
                         "s_cbranch_scc1 2f\n\t"
                         "global_load_dwordx2 %0, %4, %6\n\t"
                         "global_load_dwordx4 %1, %5, %7\n\t"
                         "s_cmp_eq_u32 %9, 0\n\t"
                         "s_cbranch_scc1 2f\n\t"
                         "s_cmp_eq_u32 %9, 1\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %3, %5, %8\n\t"
                         "s_branch 2f\n"
                         "1:\n\t"
                         "global_load_dwordx2 %2, %4, %8\n"
                         "2:"
                         : "=&v"(wt[t]), "=&v"(xa[t]), "=&v"(yr[t]), "=&v"(xb[t])     // early clobber: later instructions of the statement still read %4, %5
                         : "v"(off8), "v"(off16), "s"(pw), "s"(px), "s"(py), "s"(mode)
                         : "memory", "scc");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(wt[0]), "+v"(yr[0]), "+v"(xa[0]), "+v"(xb[0]), "+v"(wt[1]), "+v"(yr[1]), "+v"(xa[1]), "+v"(xb[1]),
                          "+v"(wt[2]), "+v"(yr[2]), "+v"(xa[2]), "+v"(xb[2]), "+v"(wt[3]), "+v"(yr[3]), "+v"(xa[3]), "+v"(xb[3]));
#pragma unroll
        for (int t = 0; t < GAIN_T; ++t) {
            if (t < n) {
                const double x0 = xa[t].x, x1 = xa[t].y;
                if (is_rhs) { const double rr = wt[t] * yr[t]; g00 += x0 * rr; g01 += x1 * rr; }
                else {
                    const bool same = cur[6 + 3 * t] == cur[5 + 3 * t];
                    const double y0 = same ? x0 : xb[t].x, y1 = same ? x1 : xb[t].y;
                    const double at = wt[t] * x0, av = wt[t] * x1;
                    g00 += at * y0; g01 += at * y1; g10 += av * y0; g11 += av * y1;
                }
            }
        }
        if (head & 256) {                                                // last record of the item
            if (is_rhs) {
                if (head & 16) g00 = 0.0;                                // the slack angle has no equation (:899)
                jg::store_vec(a.rhs, (size_t)cur[1], b, ld, g00, g01);
            } else {
                if (head & 16) { g00 = 0.0; g01 = 0.0; }                 // removeColumn(H, slack) on both sides (:885)
                if (head & 32) { g00 = 0.0; g10 = 0.0; }
                if ((head & 48) == 48) g00 = 1.0;                        // gain[slack, slack] = 1 (:889)
                jg::store_blk(a.Gv, (size_t)cur[1], b, ld, g00, g01, g10, g11);
            }
            g00 = 0.0; g01 = 0.0; g10 = 0.0; g11 = 0.0;
        }
        cur = nxt;
    }
}

constexpr int NORM_ROWS = 64;

// Correction pass of the orthogonal (Q-less) method: rho = res - H * inc over the slots of a row (the slack angle column
// of H is removed, acStateEstimation.jl:914).  One wave per row x 64 scenarios.
__global__ __launch_bounds__(256) void k_gn_hdelta(const RowDesc* rows, const int* slot_bus, const double* Hs, const double* inc,
                                                   const double* res, double* rho, int m, int slack, int ld_) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)ld_;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * GN_ROWS, r1 = min(r0 + GN_ROWS, m);
    typedef const int __attribute__((address_space(4)))* CInt;
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef const i4 __attribute__((address_space(4)))* CInt4;
    for (int r = r0 + wave; r < r1; r += blockDim.y) {
        const i4 rd = ((CInt4)rows)[r];                             // {type, idx, slot0, nslots} through the scalar cache
        const int s0 = rd[2], ns = rd[3];
        double acc = 0.0;
        for (int s = s0; s < s0 + ns; ++s) {
            const int bus = ((CInt)slot_bus)[s];
            const double2 hv = jg::load_vec(Hs, (size_t)s, b, ld), d = jg::load_vec(inc, (size_t)bus, b, ld);
            acc += (bus == slack ? 0.0 : hv.x * d.x) + hv.y * d.y;
        }
        rho[(size_t)r * ld + b] = res[(size_t)r * ld + b] - acc;
    }
}

// inc += cor (the slack angle stays whatever k_gn_norm makes of it afterwards)
__global__ __launch_bounds__(256) void k_gn_add(double* inc, const double* cor, int n, int ld) {
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * NORM_ROWS, r1 = min(r0 + NORM_ROWS, n);
    for (int i = r0 + wave; i < r1; i += 4) {
        const double2 x = jg::load_vec(inc, (size_t)i, b, ld), y = jg::load_vec(cor, (size_t)i, b, ld);
        jg::store_vec(inc, (size_t)i, b, ld, x.x + y.x, x.y + y.y);
    }
}

// ---- Monte-Carlo realisations drawn ON the device (jgrid.h: jg_gn_set_readings, jg_gn_draw_noise) -----------------------------------------------
// The reference draws a realisation inside add<Meter>!(...; noise = true): mean + variance^(1/2) * randn (src/measurement/utility.jl:70-73), and acWLS
// turns the raw readings into se.mean / se.precision row by row (acStateEstimation.jl:135-236: squared currents mean z^2, variance 4 z^2 sigma^2; rectangular
// PMUs through variancePmu / covariancePmu + precision!, equations.jl:576-666).  Drawn on the host, 512 realisations of config 4 are 50 M normals and
// 0.8 GB over PCIe per job; here a wave owns one DEVICE (meter or PMU), its lanes the realisations: two normals per (device, realisation) from a
// counter-based generator (no state: realisation r of seed s is the same numbers on any rank, in any batch, at any lane), the value rules, coalesced rows out.
//   c = (seed + GOLDEN * (2 d)) ^ (realisation * 0xD1B54A32D192ED03);  u1 = mix64(c), u2 = mix64(c + GOLDEN), mix64 = the splitmix64 finaliser;  uniform (0, 1] = ((u >> 11) + 1) 2^-53;
//   Box-Muller: e1 = sqrt(-2 ln u1) cos(2 pi u2), e2 = ... sin(...).   tests/test_montecarlo_gpu.py restates it in numpy.
struct NoiseDev { int row, kind, corr, st; double z1, v1, z2, v2; };   // kind 0 plain, 1 squared, 2 polar PMU, 3 polar PMU with squared magnitude, 4 / 5 rectangular PMU (un)correlated; st = st1 | st2 << 1
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
__global__ __launch_bounds__(1024) void k_gn_noise(const NoiseDev* dev, int ndev, unsigned long long seed, double scale, long long first, double* mean, double* w,
                                                   int m, int ld, int lanes, int* bad) {
    const int lane = threadIdx.x, wave = uniform(threadIdx.y);
    const int d = blockIdx.x * 16 + wave;
    if (d >= ndev) return;
    const size_t b = (size_t)min((int)blockIdx.y * 64 + lane, lanes - 1);
    const NoiseDev q = dev[d];
    const unsigned long long ctr = (seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * d)) ^ ((unsigned long long)(first + (long long)b) * 0xD1B54A32D192ED03ull);
    const double u1 = (double)((mix64(ctr) >> 11) + 1ull) * 0x1p-53, u2 = (double)((mix64(ctr + 0x9E3779B97F4A7C15ull) >> 11) + 1ull) * 0x1p-53;
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    const double zz = q.z1 + scale * sqrt(q.v1) * (rad * cs);
    const double za = q.z2 + scale * sqrt(q.v2) * (rad * sn);
    const double st1 = (double)(q.st & 1), st2 = (double)((q.st >> 1) & 1);
    double m0, w0, m1 = 0.0, w1 = 1.0, off = 0.0;
    if (q.kind == 0 || q.kind == 2) { m0 = st1 * zz; w0 = 1.0 / q.v1; }
    else if (q.kind == 1 || q.kind == 3) { m0 = st1 * zz * zz; w0 = 1.0 / (4.0 * zz * zz * q.v1); }
    else {
        double s, c;
        sincos(za, &s, &c);
        const double vre = q.v1 * c * c + q.v2 * (zz * s) * (zz * s), vim = q.v1 * s * s + q.v2 * (zz * c) * (zz * c);
        const double stt = st1 * st2;
        m0 = stt * (zz * c); m1 = stt * (zz * s);
        if (q.kind == 4) { w0 = 1.0 / vre; w1 = 1.0 / vim; }
        else {                                                    // covariancePmu + precision! (equations.jl:591-666)
            const double l1i = 1.0 / sqrt(vre), l2 = s * c * (q.v1 - q.v2 * zz * zz) * l1i, l3i2 = 1.0 / (vim - l2 * l2);
            off = (-l2 * l1i) * l3i2;
            w0 = (l1i - l2 * off) * l1i; w1 = l3i2;
        }
    }
    if (q.kind == 2 || q.kind == 3) { m1 = st2 * za; w1 = 1.0 / q.v2; }
    const size_t col = (size_t)blockIdx.y * 64 + lane;            // every lane of the padded batch gets a value (lanes beyond the batch repeat its last realisation)
    mean[(size_t)q.row * ld + col] = m0; w[(size_t)q.row * ld + col] = w0;
    bool ok = w0 > 0.0 && w0 < 1.0e300;
    if (q.kind >= 2) { mean[(size_t)(q.row + 1) * ld + col] = m1; w[(size_t)(q.row + 1) * ld + col] = w1; ok = ok && w1 > 0.0 && w1 < 1.0e300; }
    if (q.kind == 5) { w[(size_t)(m + q.corr) * ld + col] = off; ok = ok && off == off && fabs(off) < 1.0e300; }
    if (!ok) atomicOr(bad, 1);                                    // errorVariance: a zero-magnitude squared current / rectangular PMU reading
}

// ---- the result record of a sharded Monte-Carlo run (jgrid.h: jg_gn_pack_results_device) --------------------------------------------------
// se.objective = r' W r at the residual of the last increment! (equations.jl:689-698), per scenario, reduced where the residuals are: a noisy
// realisation's residual is 96 723 doubles, its objective one.  Two launches with a FIXED summation order (bitwise run-to-run): a wave sums OBJ_WROWS
// consecutive rows in row order, the sixteen waves of a workgroup meet in LDS in wave order, the chunks are added in chunk order, then the cross
// terms 2 r_a r_b W_ab of the correlated PMU pairs in pair order.
constexpr int OBJ_WROWS = 128, OBJ_ROWS = 16 * OBJ_WROWS;
__global__ __launch_bounds__(1024) void k_gn_obj_partial(const double* res, const double* w, double* part, int m, int ld, int lanes) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = (size_t)min((int)blockIdx.y * 64 + lane, lanes - 1);
    const int r0 = blockIdx.x * OBJ_ROWS + wave * OBJ_WROWS, r1 = min(r0 + OBJ_WROWS, m);
    double acc = 0.0;
    for (int r = r0; r < r1; r += 4) {                                   // four rows in flight, added in row order
        double x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = min(r + u, m - 1); x[u] = res[(size_t)q * ld + b]; y[u] = w[(size_t)q * ld + b]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (r + u < r1) acc += x[u] * x[u] * y[u];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        for (int k = 1; k < 16; ++k) acc += red[k][lane];
        part[(size_t)blockIdx.x * ld + blockIdx.y * 64 + lane] = acc;
    }
}
__global__ __launch_bounds__(64) void k_gn_obj_final(const double* part, int chunks, const double* res, const double* woff, const int* corr_row, int ncorr,
                                                     double* obj, int ld, int lanes) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= lanes) return;
    double acc = 0.0;
    for (int c = 0; c < chunks; ++c) acc += part[(size_t)c * ld + b];
    for (int q = 0; q < ncorr; ++q) {                                    // the cross term sits on the pair's second row in the reference (:694-698)
        const int r = corr_row[q];
        acc += 2.0 * res[(size_t)r * ld + b] * res[(size_t)(r + 1) * ld + b] * woff[(size_t)q * ld + b];
    }
    obj[b] = acc;
}
// [n][ld] rows -> the scenario-major record (the transpose kernel of the power-flow record, jg_nr.hip: k_pack_bus): z = 0 magnitudes, 1 angles
__global__ __launch_bounds__(512) void k_gn_pack_bus(const double* vm, const double* va, double* dst, int n, int ld, int batch, long long stride) {
    __shared__ double tile[64][65];
    const double* src = blockIdx.z ? va : vm;
    const int off = blockIdx.z ? n : 0;
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && b < ld) ? src[(size_t)i * ld + b] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = b0 + r, i = i0 + threadIdx.x;
        if (b < batch && i < n) dst[(size_t)b * stride + off + i] = tile[threadIdx.x][r];
    }
}
__global__ void k_gn_pack_tail(const int* iters, const int* status, const double* obj, double* dst, int batch, long long stride, int off) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) { double* q = dst + (size_t)b * stride + off; q[0] = (double)iters[b]; q[1] = (double)status[b]; q[2] = obj[b]; }
}

// max |increment| per scenario (partial per bus chunk); forces increment[slack theta] = 0 (:899)
__global__ __launch_bounds__(256) void k_gn_norm(double* inc, double* part, int n, int slack, int ld) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * NORM_ROWS, r1 = min(r0 + NORM_ROWS, n);
    double mx = 0.0;
    for (int i = r0 + wave; i < r1; i += 4) {
        const double2 iv = jg::load_vec(inc, (size_t)i, b, ld);
        double t = iv.x;
        const double v = iv.y;
        if (i == slack) { t = 0.0; jg::store_vec(inc, (size_t)i, b, ld, 0.0, v); }
        const double at = fabs(t), av = fabs(v);
        mx = (at > mx || at != at) ? at : mx;
        mx = (av > mx || av != av) ? av : mx;
    }
    red[wave][lane] = mx;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w) { const double x = red[w][lane]; mx = (x > mx || x != x) ? x : mx; }
        part[(size_t)blockIdx.x * ld + b] = mx;
    }
}

// Largest normalised residual test (src/stateEstimation/badData.jl:181-311): c_r = h_r Z h_r' from the selected inverse
// Z of the gain matrix (rowProjection, :289-311), nres_r = |res_r| / sqrt(|1 / W_rr - c_r|), 0 for rows without residual
// or without weight (removed for this scenario).  One wave per row; the pair list of a row names (slot, slot, Z entry).
struct ProjArgs {
    const int* pair_ptr; const int* pa; const int* pb; const int* pz;
    const int* slot_bus; const double* Hs; const double* Z; const double* res; const double* w;
    double* nres; int m; int slack; int ld;
};

__global__ __launch_bounds__(256) void k_gn_project(ProjArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    typedef const int __attribute__((address_space(4)))* CInt;
    CInt pp = (CInt)a.pair_ptr, pa = (CInt)a.pa, pb = (CInt)a.pb, pz = (CInt)a.pz, sb = (CInt)a.slot_bus;
    const int r0 = blockIdx.x * GN_ROWS, r1 = min(r0 + GN_ROWS, a.m);
    for (int r = r0 + wave; r < r1; r += blockDim.y) {
        double c = 0.0;
        for (int q = pp[r], e = pp[r + 1]; q < e; ++q) {
            const int sa = pa[q], sc = pb[q];
            double2 ha = jg::load_vec(a.Hs, (size_t)sa, b, ld), hb = jg::load_vec(a.Hs, (size_t)sc, b, ld);
            const jg::Blk z = jg::load_blk(a.Z, (size_t)pz[q], b, ld);
            if (sb[sa] == a.slack) ha.x = 0.0;                   // removeColumn(jacobian, slack) (:200)
            if (sb[sc] == a.slack) hb.x = 0.0;
            const double t = ha.x * (z.v00 * hb.x + z.v01 * hb.y) + ha.y * (z.v10 * hb.x + z.v11 * hb.y);
            c += sa == sc ? t : 2.0 * t;                         // Z is symmetric: the (b, a) pair is the same number
        }
        const double res = a.res[(size_t)r * ld + b], w = a.w[(size_t)r * ld + b];
        a.nres[(size_t)r * ld + b] = (res != 0.0 && w != 0.0) ? fabs(res) / sqrt(fabs(1.0 / w - c)) : 0.0;
    }
}

// (max, first argmax) of nres over the rows, per scenario: stage 1 per row chunk, stage 2 over the chunks
__global__ __launch_bounds__(256) void k_gn_argmax(const double* nres, double* part_v, int* part_i, int m, int rows_per, int ld) {
    __shared__ double rv[4][64];
    __shared__ int rix[4][64];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int r0 = blockIdx.x * rows_per, r1 = min(r0 + rows_per, m);
    const int per = (rows_per + 3) / 4;
    double mx = 0.0; int ix = -1;
    for (int r = r0 + wave * per, e = min(r + per, r1); r < e; ++r) {      // contiguous rows per wave: first maximum wins
        const double v = nres[(size_t)r * ld + b];
        if (v > mx) { mx = v; ix = r; }
    }
    rv[wave][lane] = mx; rix[wave][lane] = ix;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w) if (rv[w][lane] > mx) { mx = rv[w][lane]; ix = rix[w][lane]; }
        part_v[(size_t)blockIdx.x * ld + b] = mx; part_i[(size_t)blockIdx.x * ld + b] = ix;
    }
}

__global__ void k_gn_argmax2(const double* part_v, const int* part_i, double* out_v, int* out_i, int nchunk, int ld) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ld) return;
    double mx = 0.0; int ix = -1;
    for (int c = 0; c < nchunk; ++c) { const double v = part_v[(size_t)c * ld + b]; if (v > mx) { mx = v; ix = part_i[(size_t)c * ld + b]; } }
    out_v[b] = mx; out_i[b] = ix + 1;                            // 1-based row, 0 = every normalised residual is zero
}

struct GnCheckArgs {
    const double* part; int nchunk; int ld; int batch; const double* params;
    double* maxinc; int* active; int* iters; int* status; const int* lu_status; int* counter; int* group; int mode;
};

__global__ void k_gn_check(GnCheckArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.ld) return;
    double mx = 0.0;
    for (int c = 0; c < a.nchunk; ++c) { const double x = a.part[(size_t)c * a.ld + b]; mx = (x > mx || x != x) ? x : mx; }
    a.maxinc[b] = mx;
    if (a.mode == 0) return;
    const double tol = a.params[0];
    const int maxit = (int)a.params[1];
    const bool conv = mx < tol;                                   // acStateEstimation.jl:1307
    const bool bad = (a.lu_status[b] & 4) || mx != mx;
    const bool act = b < a.batch && !conv && !bad && a.iters[b] < maxit;   // :1311
    a.active[b] = act ? 1 : 0;
    a.status[b] = conv ? 0 : (bad ? 3 : 1);
    if (act) { a.iters[b] += 1; atomicAdd(a.counter, 1); atomicOr(a.group + (b >> 6), 1); }
}

// solve!: theta += inc[1:n], V += inc[n+1:2n] on active scenarios (:1040-1043)
__global__ __launch_bounds__(256) void k_gn_update(const double* inc, double* vm, double* va, const int* active, int n, int ld) {
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    if (active && !active[b]) return;
    const int r0 = blockIdx.x * NORM_ROWS, r1 = min(r0 + NORM_ROWS, n);
    for (int i = r0 + wave; i < r1; i += 4) {
        const double2 iv = jg::load_vec(inc, (size_t)i, b, ld);
        va[(size_t)i * ld + b] += iv.x;
        vm[(size_t)i * ld + b] += iv.y;
    }
}

__global__ void k_gn_add_iter(int* iters, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) iters[b] += 1;
}

}  // namespace

struct jg_gn {
    int n = 0, nnzY = 0, nb = 0, m = 0, batch = 0, ld = 0, device = 0, slack0 = 0, nslots = 0, ncorr = 0, nchunk = 0;
    int64_t nnzH = 0;
    std::vector<int64_t> hcolptr, hrowval;      // reference CSC pattern of H (1-based)
    std::vector<int64_t> hmap;                  // CSC nz -> slot*2 + comp
    std::vector<int> gi_rowptr, gi_col;         // gain block CSR
    std::vector<int8_t> type, code;
    std::vector<int> corr_row;
    int gain_waves = 0, rhs_waves = 0;          // waves of the gain launch / of the rhs-only (correction) launch
    // device
    RowDesc* d_rows = nullptr; int* d_slot_bus = nullptr; BranchP* d_br = nullptr; int* d_items = nullptr; int n_items = 0;
    int* d_rowptr = nullptr; double* d_G = nullptr; double* d_B = nullptr; int* d_ydiag = nullptr;
    void* d_arena = nullptr;                    // one allocation behind the per-handle state below (jg_gn_create)
    double* d_vm = nullptr; double* d_va = nullptr; double* d_mean = nullptr; double* d_w = nullptr;
    double* d_Hs = nullptr; double* d_res = nullptr; double* d_rhs = nullptr; double* d_inc = nullptr;
    GainRec* d_grec = nullptr; int* d_gwave = nullptr; GainRec* d_rrec = nullptr; int* d_rwave = nullptr;   // gain + rhs records of the items that are NOT staged (k_gn_gain), rhs records alone
    // bad-data test (built on first use)
    std::vector<RowDesc> rows_host; std::vector<int> slot_bus_host;
    int* d_pair_ptr = nullptr; int* d_pa = nullptr; int* d_pb = nullptr; int* d_pz = nullptr;
    double* d_nres = nullptr; double* d_amax_v = nullptr; int* d_amax_i = nullptr; double* d_bad_v = nullptr; int* d_bad_i = nullptr;
    int amax_chunks = 0;
    double* d_part = nullptr; double* d_maxinc = nullptr; double* d_params = nullptr;
    double* d_vm0 = nullptr; double* d_va0 = nullptr;   // snapshot of the start point (jg_gn_snapshot_voltage)
    int method = 0;                                     // jg_gn_set_method: 0 normal equations, 1 + one least-squares correction pass
    double* d_rho = nullptr; double* d_rhs2 = nullptr; double* d_inc2 = nullptr;   // correction pass (allocated by jg_gn_set_method)
    int* d_active = nullptr; int* d_iters = nullptr; int* d_status = nullptr; int* d_counter = nullptr; int* d_group = nullptr;
    NoiseDev* d_noise = nullptr; int n_noise = 0; int* d_noise_bad = nullptr;   // raw readings per device (jg_gn_set_readings)
    double* d_obj = nullptr; double* d_objpart = nullptr; int* d_corr = nullptr; int obj_chunks = 0;   // objective per scenario (first use: jg_gn_get_objective / jg_gn_pack_results_device)
    bool ran = false;                                   // d_iters / d_status hold the verdicts of a stateEstimation! run
    double wait_us = 0.0;                               // running mean of the host's waits for an iteration's verdict (jg_gn_run: polls while this is short)
    double* d_stage = nullptr; size_t stage_bytes = 0;  // rows on their way up or down (put_rows / get_rows)
    jg::Engine eng;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    int* h_counter = nullptr;
};

namespace {

int set_device(jg_gn* h) { GN_HIP(hipSetDevice(h->device)); return 0; }

void launch_rows(jg_gn* h) {
    RowArgs a{h->d_rows, h->d_slot_bus, h->d_br, h->d_rowptr, h->d_G, h->d_B, h->d_ydiag, h->d_vm, h->d_va, h->d_mean,
              h->d_Hs, h->d_res, h->m, h->ld, h->d_items, h->n_items};
    hipLaunchKernelGGL(k_gn_rows, dim3((h->n_items + GN_ITEMS - 1) / GN_ITEMS, h->ld / 64), dim3(64, 4), 0, h->stream, a);
}

void launch_gain(jg_gn* h, bool correction = false) {
    const int nw = correction ? h->rhs_waves : h->gain_waves;
    GainArgs a{correction ? h->d_rrec : h->d_grec, correction ? h->d_rwave : h->d_gwave, h->d_Hs, correction ? h->d_rho : h->d_res, h->d_w, h->eng.X,
               correction ? h->d_rhs2 : h->d_rhs, nw, h->ld};
    if (nw > 0) hipLaunchKernelGGL(k_gn_gain, dim3(jg::grid_blocks(h->ld / 64, (nw + 3) / 4)), dim3(64, 4), 0, h->stream, a);
}

int launch_increment(jg_gn* h, const int* group) {
    launch_rows(h);
    launch_gain(h);
    if (int rc = h->eng.factor(h->stream, nullptr, h->d_rhs, jg::GroupSel{group})) return failg(rc, h->eng.error);
    jg::StateUpdate none{};
    if (int rc = h->eng.backsolve(h->stream, h->d_inc, none, jg::GroupSel{group})) return failg(rc, h->eng.error);
    if (h->method == 1) {
        // Orthogonal / Peters-Wilkinson tags (acStateEstimation.jl:906-971): the increment is the least-squares solution of
        // sqrt(W) H d = sqrt(W) r.  The triangular factor R of the reference's QR IS the Cholesky factor of the gain the engine
        // already holds (R'R = H'WH); Q is never formed.  One correction pass with the residual taken through H itself
        // (rho = r - H d, d += G^-1 H'W rho: the corrected semi-normal equations) gives the increment the error bound of the QR
        // method instead of the squared condition number of the plain normal equations.
        hipLaunchKernelGGL(k_gn_hdelta, dim3((h->m + GN_ROWS - 1) / GN_ROWS, h->ld / 64), dim3(64, 4), 0, h->stream, h->d_rows, h->d_slot_bus, h->d_Hs,
                           h->d_inc, h->d_res, h->d_rho, h->m, h->slack0, h->ld);
        launch_gain(h, true);
        if (int rc = h->eng.forward(h->stream, h->d_rhs2, jg::GroupSel{group})) return failg(rc, h->eng.error);
        if (int rc = h->eng.backsolve(h->stream, h->d_inc2, none, jg::GroupSel{group})) return failg(rc, h->eng.error);
        hipLaunchKernelGGL(k_gn_add, dim3(h->nchunk, h->ld / 64), dim3(64, 4), 0, h->stream, h->d_inc, h->d_inc2, h->n, h->ld);
    }
    hipLaunchKernelGGL(k_gn_norm, dim3(h->nchunk, h->ld / 64), dim3(64, 4), 0, h->stream, h->d_inc, h->d_part, h->n, h->slack0, h->ld);
    return 0;
}

void launch_check(jg_gn* h, int mode) {
    GnCheckArgs c{h->d_part, h->nchunk, h->ld, h->batch, h->d_params, h->d_maxinc, h->d_active, h->d_iters, h->d_status,
                  h->eng.status, h->d_counter, h->d_group, mode};
    hipLaunchKernelGGL(k_gn_check, dim3((h->ld + 255) / 256), dim3(256), 0, h->stream, c);
}

void launch_update(jg_gn* h, const int* active) {
    hipLaunchKernelGGL(k_gn_update, dim3(h->nchunk, h->ld / 64), dim3(64, 4), 0, h->stream, h->d_inc, h->d_vm, h->d_va, active, h->n, h->ld);
}

// host [batch][rows] (stride = rows) or ONE [rows] for every scenario (stride 0) -> device [rows][ld]; lanes beyond the batch repeat the last scenario.
// The rows go up as the host holds them and a kernel spreads them over the lanes (round 5: the host-side transposition that used to happen here wrote
// 96 723 x 512 doubles with a stride of 4 KB and uploaded 396 MB even for one shared column -- 0.95 of the 1.4 s a Gauss-Newton handle of config 4 took to build).
__global__ __launch_bounds__(512) void k_gn_spread_rows(const double* src, double* dst, int rows, int ld, int batch, int src_rows) {
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = min(min(b0 + r, batch - 1), src_rows - 1), i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = i < rows ? src[(size_t)b * rows + i] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        if (i < rows && b < ld) dst[(size_t)i * ld + b] = tile[threadIdx.x][r];
    }
}
// device [rows][ld] -> device [batch][rows] (what the host receives)
__global__ __launch_bounds__(512) void k_gn_collect_rows(const double* src, double* dst, int rows, int ld, int batch) {
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < rows && b < ld) ? src[(size_t)i * ld + b] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = b0 + r, i = i0 + threadIdx.x;
        if (b < batch && i < rows) dst[(size_t)b * rows + i] = tile[threadIdx.x][r];
    }
}

// the staging area of a handle for rows on their way up or down: grows, is never freed before the handle (a hipFree per call would synchronise the device -- while
// another host thread of a pipeline may be capturing its hipGraph)
int stage_room(jg_gn* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return 0;
    GN_HIP(hipStreamSynchronize(h->stream));
    hipFree(h->d_stage); h->d_stage = nullptr; h->stage_bytes = 0;
    GN_HIP(hipMalloc((void**)&h->d_stage, bytes));
    h->stage_bytes = bytes;
    return 0;
}

int put_rows(jg_gn* h, double* dst, const double* src, int64_t stride, int rows) {
    if (stride != 0 && stride != rows) {                         // a caller's own row pitch: the general (slow) way
        std::vector<double> t((size_t)rows * h->ld, 0.0);
        for (int b = 0; b < h->ld; ++b) {
            const double* s = src + (size_t)(b < h->batch ? b : h->batch - 1) * (size_t)stride;
            for (int i = 0; i < rows; ++i) t[(size_t)i * h->ld + b] = s[i];
        }
        GN_HIP(jg::sync_copy(dst, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    const int src_rows = stride == 0 ? 1 : h->batch;
    const size_t bytes = (size_t)src_rows * rows * sizeof(double);
    if (int rc = stage_room(h, bytes)) return rc;
    GN_HIP(jg::sync_copy(h->d_stage, src, bytes, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_gn_spread_rows, dim3((rows + 63) / 64, h->ld / 64), dim3(64, 8), 0, h->stream, (const double*)h->d_stage, dst, rows, h->ld, h->batch, src_rows);
    GN_HIP(hipGetLastError());
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int get_rows(jg_gn* h, const double* src, double* dst, size_t rows) {
    const size_t bytes = (size_t)h->batch * rows * sizeof(double);
    if (int rc = stage_room(h, bytes)) return rc;
    hipLaunchKernelGGL(k_gn_collect_rows, dim3((unsigned)((rows + 63) / 64), h->ld / 64), dim3(64, 8), 0, h->stream, src, h->d_stage, (int)rows, h->ld, h->batch);
    GN_HIP(hipGetLastError());
    GN_HIP(jg::sync_copy(dst, h->d_stage, bytes, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

}  // namespace

extern "C" {

int jg_gn_create(jg_gn** out, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* y_reim,
                 const double* yt_reim, int64_t nb, const int64_t* from, const int64_t* to, const double* branch_param,
                 int64_t slack, int64_t m, const int8_t* code, const int8_t* status, const int64_t* index,
                 int64_t n_corr, const int64_t* corr_row, int64_t batch, int device) {
    if (!out || n < 1 || !colptr || !rowval || !y_reim || !yt_reim || nb < 0 || m < 1 || !code || !status || !index ||
        batch < 1 || slack < 0 || slack > n || n_corr < 0 || (n_corr > 0 && !corr_row) || (nb > 0 && (!from || !to || !branch_param)))
        return failg(1, "jg_gn_create: bad argument");
    int ndev = 0;
    GN_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return failg(1, "jg_gn_create: no such HIP device");
    jg_gn* h = new jg_gn();
    h->n = (int)n; h->nb = (int)nb; h->m = (int)m; h->batch = (int)batch; h->ld = (int)((batch + 63) / 64 * 64);
    h->device = device; h->slack0 = (int)slack - 1; h->ncorr = (int)n_corr;
    h->nnzY = (int)(colptr[n] - 1);
    const int nnzY = h->nnzY;
    // Ybus CSR rows (= transposed CSC, symmetric pattern) and diagonal pointers
    std::vector<int> rp(n + 1), cl(nnzY), ydiag(n, -1);
    std::vector<double> G(nnzY), B(nnzY);
    for (int i = 0; i <= n; ++i) rp[i] = (int)(colptr[i] - 1);
    for (int p = 0; p < nnzY; ++p) { cl[p] = (int)(rowval[p] - 1); G[p] = yt_reim[2 * p]; B[p] = yt_reim[2 * p + 1]; }
    for (int i = 0; i < n; ++i)
        for (int p = rp[i]; p < rp[i + 1]; ++p) if (cl[p] == i) ydiag[i] = p;
    for (int i = 0; i < n; ++i) if (ydiag[i] < 0) { delete h; return failg(1, "jg_gn_create: Ybus diagonal missing"); }
    (void)y_reim;
    // ---- rows, slots and the reference's H pattern (acWLS + index builders, :135-238, :1130-1238) ----
    std::vector<RowDesc> rows(m);
    std::vector<int> slot_bus;
    h->type.assign(m, 0);
    h->code.assign(code, code + m);
    struct Trip { int64_t row, col; int slotcomp; };
    std::vector<Trip> trips;
    for (int64_t r = 0; r < m; ++r) {
        const int cd = code[r];
        const int st = status[r];
        if (st != 0 && st != 1) { delete h; return failg(1, "jg_gn_create: status must be 0 or 1"); }
        const int64_t k = index[r] - 1;
        RowDesc d{(int)(st * cd), (int)k, (int)slot_bus.size(), 0};
        h->type[r] = (int8_t)(st * cd);
        const int s0 = (int)slot_bus.size();
        if (cd == 1 || cd == 12 || cd == 13 || cd == 16 || cd == 17) {
            if (k < 0 || k >= n) { delete h; return failg(1, "jg_gn_create: bus index out of range"); }
            slot_bus.push_back((int)k);
            if (cd == 13) trips.push_back({r, k, s0 * 2});                       // oneIndices!(..., col = k)
            else if (cd == 1 || cd == 12) trips.push_back({r, k + n, s0 * 2 + 1}); // oneIndices!(..., col = k + n)
            else { trips.push_back({r, k, s0 * 2}); trips.push_back({r, k + n, s0 * 2 + 1}); }   // twoIndices!
        } else if (cd == 6 || cd == 9) {                                         // nthIndices!
            if (k < 0 || k >= n) { delete h; return failg(1, "jg_gn_create: bus index out of range"); }
            for (int p = rp[k]; p < rp[k + 1]; ++p) {
                const int s = (int)slot_bus.size();
                slot_bus.push_back(cl[p]);
                trips.push_back({r, (int64_t)cl[p], s * 2}); trips.push_back({r, (int64_t)cl[p] + n, s * 2 + 1});
            }
        } else if ((cd >= 2 && cd <= 5) || cd == 7 || cd == 8 || cd == 10 || cd == 11 || cd == 14 || cd == 15 || (cd >= 18 && cd <= 21)) {
            if (k < 0 || k >= nb) { delete h; return failg(1, "jg_gn_create: branch index out of range"); }   // fourIndices!
            const int64_t f = from[k] - 1, t = to[k] - 1;
            slot_bus.push_back((int)f); slot_bus.push_back((int)t);
            trips.push_back({r, f, s0 * 2}); trips.push_back({r, t, (s0 + 1) * 2});
            trips.push_back({r, f + n, s0 * 2 + 1}); trips.push_back({r, t + n, (s0 + 1) * 2 + 1});
        } else if (cd == 22 || cd == 23) {                                       // linear PMU model, bus phasor: pmuIndices (pmuStateEstimation.jl:549-557)
            if (k < 0 || k >= n) { delete h; return failg(1, "jg_gn_create: bus index out of range"); }
            slot_bus.push_back((int)k);
            if (cd == 22) trips.push_back({r, k, s0 * 2}); else trips.push_back({r, k + n, s0 * 2 + 1});
        } else if (cd >= 24 && cd <= 27) {                                       // linear PMU model, branch current phasor
            if (k < 0 || k >= nb) { delete h; return failg(1, "jg_gn_create: branch index out of range"); }
            const int64_t f = from[k] - 1, t = to[k] - 1;
            slot_bus.push_back((int)f); slot_bus.push_back((int)t);
            trips.push_back({r, f, s0 * 2}); trips.push_back({r, t, (s0 + 1) * 2});
            trips.push_back({r, f + n, s0 * 2 + 1}); trips.push_back({r, t + n, (s0 + 1) * 2 + 1});
        } else { delete h; return failg(1, "jg_gn_create: unknown measurement type code"); }
        d.nslots = (int)slot_bus.size() - s0;
        rows[r] = d;
    }
    h->nslots = (int)slot_bus.size();
    h->rows_host = rows; h->slot_bus_host = slot_bus;
    // sparse(row, col, val, m, 2n): CSC, rows ascending inside a column (rows are generated ascending)
    h->hcolptr.assign(2 * n + 1, 0);
    for (const Trip& t : trips) h->hcolptr[t.col + 1]++;
    h->hcolptr[0] = 1;
    for (int64_t c = 0; c < 2 * n; ++c) h->hcolptr[c + 1] += h->hcolptr[c];
    h->nnzH = h->hcolptr[2 * n] - 1;
    h->hrowval.assign(h->nnzH, 0); h->hmap.assign(h->nnzH, 0);
    {
        std::vector<int64_t> fill(h->hcolptr.begin(), h->hcolptr.end() - 1);
        for (const Trip& t : trips) { const int64_t p = fill[t.col]++ - 1; h->hrowval[p] = t.row + 1; h->hmap[p] = t.slotcomp; }
        for (int64_t c = 0; c < 2 * n; ++c)
            for (int64_t p = h->hcolptr[c]; p < h->hcolptr[c + 1] - 1; ++p)
                if (h->hrowval[p] <= h->hrowval[p - 1]) { delete h; return failg(1, "jg_gn_create: duplicate H entry"); }
    }
    // ---- gain pattern and gather lists -------------------------------------------------------------
    h->corr_row.assign(n_corr, 0);
    std::vector<int> pair_of(m, -1);
    for (int64_t q = 0; q < n_corr; ++q) {
        const int64_t r = corr_row[q] - 1;
        if (r < 0 || r + 1 >= m) { delete h; return failg(1, "jg_gn_create: correlated pair out of range"); }
        h->corr_row[q] = (int)r; pair_of[r] = (int)q;
    }
    struct Contrib { int w, a, b; };
    std::map<std::pair<int, int>, std::vector<Contrib>> gmap;
    std::vector<std::vector<Contrib>> rmap(n);
    for (int i = 0; i < n; ++i) gmap[{i, i}];                                    // full diagonal for the engine
    for (int r = 0; r < m; ++r) {
        const RowDesc& d = rows[r];
        for (int a = 0; a < d.nslots; ++a) {
            for (int b2 = 0; b2 < d.nslots; ++b2) gmap[{slot_bus[d.slot0 + a], slot_bus[d.slot0 + b2]}].push_back({r, d.slot0 + a, d.slot0 + b2});
            rmap[slot_bus[d.slot0 + a]].push_back({r, d.slot0 + a, r});
        }
        if (pair_of[r] >= 0) {                                                    // 2x2 precision block of a correlated PMU
            const RowDesc& e = rows[r + 1];
            const int wq = (int)m + pair_of[r];
            for (int a = 0; a < d.nslots; ++a)
                for (int b2 = 0; b2 < e.nslots; ++b2) {
                    gmap[{slot_bus[d.slot0 + a], slot_bus[e.slot0 + b2]}].push_back({wq, d.slot0 + a, e.slot0 + b2});
                    gmap[{slot_bus[e.slot0 + b2], slot_bus[d.slot0 + a]}].push_back({wq, e.slot0 + b2, d.slot0 + a});
                }
            for (int a = 0; a < d.nslots; ++a) rmap[slot_bus[d.slot0 + a]].push_back({wq, d.slot0 + a, r + 1});
            for (int b2 = 0; b2 < e.nslots; ++b2) rmap[slot_bus[e.slot0 + b2]].push_back({wq, e.slot0 + b2, r});
        }
    }
    // gain pattern -> symbolic analysis (the engine must exist before the gather lists: a symmetric engine reads only the
    // blocks on and above the diagonal IN PIVOT ORDER, so only those are gathered)
    std::vector<int> blk_row, blk_col;
    h->gi_rowptr.assign(n + 1, 0);
    for (const auto& kv : gmap) {
        blk_row.push_back(kv.first.first); blk_col.push_back(kv.first.second);
        h->gi_rowptr[kv.first.first + 1]++;
        h->gi_col.push_back(kv.first.second);
    }
    for (int i = 0; i < n; ++i) h->gi_rowptr[i + 1] += h->gi_rowptr[i];
    int rc = set_device(h);
    if (rc) { delete h; return rc; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return failg(2, "jg_gn_create: stream creation failed"); }
    // in place (k_gn_gain writes into the factor storage) + symmetric (the factorisation reads U(k,i)' for Lh(i,k): LDL'
    // with half the update terms)
    // bit 49: Jordan rows for the pivots of the top tasks (jg_symbolic.hpp).  What reads the in-task triangle of U -- the forward elimination
    // of the orthogonal method, the selected inverse of the bad-data test -- factorises with Engine::jordan off (below).
    rc = h->eng.create((int)n, h->gi_rowptr.data(), h->gi_col.data(), h->ld, 3 | 1LL << 49, h->stream);
    if (rc) { std::string msg = h->eng.error; jg_gn_destroy(h); return failg(rc, msg); }
    const std::vector<int>& ip = h->eng.plan->S.iperm;
    // wave records (see k_gn_gain): bus rows in PIVOT order -- the postorder of the elimination tree keeps electrical
    // neighbours together whatever the bus numbering of the case is, and neighbours are what shares measurement rows
    std::vector<GainRec> grec, rrec;
    std::vector<int> gwave{0}, rwave{0};
    {
        const int slack = h->slack0;                                             // -1: the model has no slack (PMU-only)
        const std::vector<int>& src_entry = h->eng.plan->S.src_entry;
        auto emit = [&](std::vector<GainRec>& out, int head, int dst, const std::vector<Contrib>& cs) {
            size_t q = 0;
            do {
                GainRec r{};
                r.w[0] = head; r.w[1] = dst;
                int nt = 0;
                for (; nt < GAIN_T && q < cs.size(); ++nt, ++q) { r.w[4 + 3 * nt] = cs[q].w; r.w[5 + 3 * nt] = cs[q].a; r.w[6 + 3 * nt] = cs[q].b; }
                r.w[2] = nt;
                if (q >= cs.size()) r.w[0] |= 256;
                out.push_back(r);
            } while (q < cs.size());
        };
        std::vector<std::vector<std::pair<int, const std::vector<Contrib>*>>> by_row(n);   // (block id, contributions) per bus row
        int id = 0;
        for (const auto& kv : gmap) {
            const int this_id = id++;
            if (ip[kv.first.first] > ip[kv.first.second]) continue;              // below the diagonal in pivot order: never read
            by_row[kv.first.first].push_back({this_id, &kv.second});
        }
        const std::vector<int>& perm = h->eng.plan->S.perm;
        constexpr size_t WAVE_RECS = 4;                                          // a wave's share: about 4 records, whole items (2: 1.25 ms, 4: 1.19, 8: 1.21, 32: 1.50 at 512 scenarios)
        // (a variant that staged a run of bus rows' operands in LDS fetched a third less and ran a third slower -- a dependent chain per task against 32 waves x 12 loads in
        // flight per CU: tools/experiments/r06_retired_kernels.patch, profiles/r02_gain_lds.txt)
        // (the order of the bus rows in the gather is free -- every item names its destination; pivot order, Cuthill-McKee on the
        // gain graph and on the bus graph measure the same 1.23-1.24 ms at 512 realisations, the case's own numbering 1.44)
        for (int k = 0; k < n; ++k) {
            const int i = perm[k];
            for (const auto& blk : by_row[i]) {
                const int j = blk_col[blk.first];
                const int head = 0 | (i == slack ? 16 : 0) | (j == slack ? 32 : 0);
                emit(grec, head, src_entry[blk.first], *blk.second);
                if (grec.size() - (size_t)gwave.back() >= WAVE_RECS) gwave.push_back((int)grec.size());
            }
            emit(grec, 1 | (i == slack ? 16 : 0), i, rmap[i]);
            if (grec.size() - (size_t)gwave.back() >= WAVE_RECS) gwave.push_back((int)grec.size());
            emit(rrec, 1 | (i == slack ? 16 : 0), i, rmap[i]);
            if (rrec.size() - (size_t)rwave.back() >= WAVE_RECS) rwave.push_back((int)rrec.size());
        }
        if (gwave.back() != (int)grec.size()) gwave.push_back((int)grec.size());
        if (rwave.back() != (int)rrec.size()) rwave.push_back((int)rrec.size());
        h->gain_waves = (int)gwave.size() - 1; h->rhs_waves = (int)rwave.size() - 1;
    }
    // work items of k_gn_rows: rows that share their operands go to one wave (see the kernel).  Grouped by the type CODE of a row (a row that is out of
    // service later keeps its place and leaves zeros); a group sits where its first row sat, the other rows leave their places.
    std::vector<int> items;
    {
        std::map<int, int> flow_item, inj_item;                                   // branch / bus -> its item
        std::vector<int> quad;                                                    // 4 ints per item
        constexpr bool fuse = true;                                               // (every row on its own: tools/r04_rows_fuse.sh measured the fused items faster)
        for (int r = 0; r < m; ++r) {
            const int c = h->code[r], idx = rows[r].idx;
            const bool flow = c == 7 || c == 8 || c == 10 || c == 11, inj = c == 6 || c == 9;
            if (fuse && flow) {
                auto it = flow_item.find(idx);
                if (it != flow_item.end()) {
                    int* q = &quad[(size_t)it->second * 4];
                    int k = 1;
                    while (k < 4 && q[k] >= 0) ++k;
                    if (k < 4) { q[k] = r; q[0] = (q[0] & 0x0fffffff) | 1 << 28; continue; }
                } else flow_item[idx] = (int)(quad.size() / 4);
            } else if (fuse && inj) {
                auto it = inj_item.find(idx);
                if (it != inj_item.end()) {
                    int* q = &quad[(size_t)it->second * 4];
                    const int ra = q[0] & 0x0fffffff;
                    if (q[1] < 0 && h->code[ra] != c && rows[ra].nslots == rows[r].nslots) {       // one active + one reactive row of the bus, 6 first
                        if (c == 6) { q[1] = ra; q[0] = r | 2 << 28; } else { q[1] = r; q[0] = ra | 2 << 28; }
                        continue;
                    }
                } else inj_item[idx] = (int)(quad.size() / 4);
            }
            quad.insert(quad.end(), {r, -1, -1, -1});
        }
        items.swap(quad);
        h->n_items = (int)(items.size() / 4);
    }
    // branch parameters
    std::vector<BranchP> br(std::max<int64_t>(nb, 1));
    for (int64_t k = 0; k < nb; ++k) {
        const double* p = branch_param + 6 * k;
        br[k] = BranchP{p[0], p[1], 0.5 * p[2], 0.5 * p[3], 1.0 / p[4], p[5], (int)(from[k] - 1), (int)(to[k] - 1), 0.0};
    }
    // ---- device -------------------------------------------------------------------------------------
    std::string err;
    if (jg::upload(&h->d_rows, rows, err, h->stream) || jg::upload(&h->d_items, items, err, h->stream) || jg::upload(&h->d_slot_bus, slot_bus, err, h->stream) || jg::upload(&h->d_br, br, err, h->stream) ||
        jg::upload(&h->d_rowptr, rp, err, h->stream) || jg::upload(&h->d_G, G, err, h->stream) || jg::upload(&h->d_B, B, err, h->stream) || jg::upload(&h->d_ydiag, ydiag, err, h->stream) ||
        jg::upload(&h->d_grec, grec, err, h->stream) || jg::upload(&h->d_gwave, gwave, err, h->stream) || jg::upload(&h->d_rrec, rrec, err, h->stream) ||
        jg::upload(&h->d_rwave, rwave, err, h->stream)) {
        jg_gn_destroy(h); return failg(2, err);
    }
    const size_t ld = h->ld;
    h->nchunk = (h->n + NORM_ROWS - 1) / NORM_ROWS;
    // the per-handle state as ONE allocation and ONE fill (as in jg_nr_create), every array on a 256-byte boundary.  (Where the arrays sit against each
    // other does not matter to the kernels: offsets of 0 ... 260 kB between consecutive arrays modulo 2 MiB measured the same k_gn_rows, 0.89 - 0.93 ms on
    // the estimate's state -- profiles/r04_gn_skew_probe.txt.)
    const size_t nw = (size_t)m + (size_t)std::max<int64_t>(n_corr, 0);
    struct Part { void** p; size_t bytes; };
    const Part parts[] = {
        {(void**)&h->d_vm, n * ld * 8}, {(void**)&h->d_va, n * ld * 8}, {(void**)&h->d_mean, (size_t)m * ld * 8}, {(void**)&h->d_w, nw * ld * 8},
        {(void**)&h->d_Hs, (size_t)h->nslots * 2 * ld * 8}, {(void**)&h->d_res, (size_t)m * ld * 8}, {(void**)&h->d_rhs, n * 2 * ld * 8},
        {(void**)&h->d_inc, n * 2 * ld * 8}, {(void**)&h->d_part, (size_t)h->nchunk * ld * 8}, {(void**)&h->d_maxinc, ld * 8}, {(void**)&h->d_params, 16},
        {(void**)&h->d_active, ld * 4}, {(void**)&h->d_iters, ld * 4}, {(void**)&h->d_status, ld * 4}, {(void**)&h->d_counter, 4},
        {(void**)&h->d_group, (ld / 64) * 4}};
    size_t arena_bytes = 0;
    for (const Part& q : parts) arena_bytes += (q.bytes + 255) / 256 * 256;
    bool ok = hipMalloc((void**)&h->d_arena, arena_bytes) == hipSuccess && jg::sync_fill(h->d_arena, 0, arena_bytes, h->stream) == hipSuccess;
    if (ok) {
        size_t off = 0;
        for (const Part& q : parts) { *q.p = (char*)h->d_arena + off; off += (q.bytes + 255) / 256 * 256; }
    }
    if (!ok || hipHostMalloc((void**)&h->h_counter, sizeof(int)) != hipSuccess) {
        jg_gn_destroy(h); return failg(2, "jg_gn_create: device allocation failed");
    }
    *out = h;
    return 0;
}

void jg_gn_destroy(jg_gn* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->exec) hipGraphExecDestroy(h->exec);
    if (h->graph) hipGraphDestroy(h->graph);
    h->eng.destroy();
    hipFree(h->d_rows); hipFree(h->d_items); hipFree(h->d_slot_bus); hipFree(h->d_br); hipFree(h->d_rowptr); hipFree(h->d_G); hipFree(h->d_B); hipFree(h->d_ydiag);
    hipFree(h->d_vm0); hipFree(h->d_va0); hipFree(h->d_rho); hipFree(h->d_rhs2); hipFree(h->d_inc2);
    hipFree(h->d_arena);                                         // V, theta, z, weights, slots, residual, rhs, increment, norms, lane bookkeeping: one allocation (jg_gn_create)
    hipFree(h->d_pair_ptr); hipFree(h->d_pa); hipFree(h->d_pb); hipFree(h->d_pz); hipFree(h->d_nres); hipFree(h->d_amax_v); hipFree(h->d_amax_i);
    hipFree(h->d_bad_v); hipFree(h->d_bad_i);
    hipFree(h->d_obj); hipFree(h->d_objpart); hipFree(h->d_corr); hipFree(h->d_noise); hipFree(h->d_noise_bad); hipFree(h->d_stage);
    hipFree(h->d_grec); hipFree(h->d_gwave); hipFree(h->d_rrec); hipFree(h->d_rwave);
    if (h->h_counter) hipHostFree(h->h_counter);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int jg_gn_dims(jg_gn* h, int64_t* dims) {
    if (!h || !dims) return failg(1, "jg_gn_dims: bad argument");
    dims[0] = h->m; dims[1] = h->nnzH; dims[2] = (int64_t)h->gi_col.size(); dims[3] = h->eng.plan->S.n_entries;
    dims[4] = h->eng.plan->S.n_sched_terms; dims[5] = (int64_t)(h->eng.fact.size() + h->eng.plan->S.top_launch.size()); dims[6] = (int64_t)h->eng.bwd.size();
    dims[7] = h->nslots;
    return 0;
}

int jg_gn_set_method(jg_gn* h, int method) {
    if (!h || method < 0 || method > 1) return failg(1, "jg_gn_set_method: method must be 0 (normal equations) or 1 (orthogonal: corrected semi-normal equations)");
    if (method == 1 && h->ncorr > 0) return failg(1, "jg_gn_set_method: the orthogonal method needs a diagonal precision matrix (no correlated PMUs)");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (method == 1 && !h->d_rho) {
        const size_t ld = h->ld;
        auto dmalloc = [&](void** p, size_t bytes) -> bool { return hipMalloc(p, bytes) == hipSuccess && jg::sync_fill(*p, 0, bytes, h->stream) == hipSuccess; };
        if (!dmalloc((void**)&h->d_rho, (size_t)h->m * ld * 8) || !dmalloc((void**)&h->d_rhs2, (size_t)h->n * 2 * ld * 8) || !dmalloc((void**)&h->d_inc2, (size_t)h->n * 2 * ld * 8))
            return failg(2, "jg_gn_set_method: device allocation failed");
    }
    if (method != h->method && h->exec) {               // the captured iteration holds the other launch sequence
        hipGraphExecDestroy(h->exec); hipGraphDestroy(h->graph);
        h->exec = nullptr; h->graph = nullptr;
    }
    h->method = method;
    h->eng.jordan = method == 0 && h->eng.plan->S.jordan && jg::knob("JORDAN", 1) != 0;   // method 1 runs forward() on the factor
    return 0;
}

int jg_gn_set_measurement(jg_gn* h, const double* mean, const double* wdiag, const double* woff, int64_t batch_stride_m,
                          int64_t batch_stride_corr) {
    if (!h || !mean || !wdiag || (h->ncorr > 0 && !woff)) return failg(1, "jg_gn_set_measurement: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (int rc = put_rows(h, h->d_mean, mean, batch_stride_m, h->m)) return rc;
    if (int rc = put_rows(h, h->d_w, wdiag, batch_stride_m, h->m)) return rc;
    if (h->ncorr > 0)
        if (int rc = put_rows(h, h->d_w + (size_t)h->m * h->ld, woff, batch_stride_corr, h->ncorr)) return rc;
    return 0;
}

int jg_gn_set_readings(jg_gn* h, int64_t ndev, const int64_t* row, const int8_t* kind, const double* z1, const double* v1, const int8_t* s1,
                       const double* z2, const double* v2, const int8_t* s2) {
    if (!h || ndev < 1 || !row || !kind || !z1 || !v1 || !s1 || !z2 || !v2 || !s2) return failg(1, "jg_gn_set_readings: bad argument");
    if (int rc = set_device(h)) return rc;
    std::vector<NoiseDev> t((size_t)ndev);
    std::vector<char> seen((size_t)h->m, 0);
    std::vector<int> pair_of((size_t)h->m, -1);
    for (int qn = 0; qn < h->ncorr; ++qn) pair_of[h->corr_row[qn]] = qn;
    for (int64_t d = 0; d < ndev; ++d) {
        const int k = kind[d], rows = k >= 2 ? 2 : 1;
        if (k < 0 || k > 5 || row[d] < 1 || row[d] + rows - 1 > h->m) return failg(1, "jg_gn_set_readings: kind 0..5, rows 1-based inside the model");
        if (!(v1[d] > 0.0) || (k >= 2 && !(v2[d] > 0.0))) return failg(1, "jg_gn_set_readings: variances must be positive");
        const int r = (int)row[d] - 1;
        for (int x = 0; x < rows; ++x) { if (seen[r + x]) return failg(1, "jg_gn_set_readings: two devices claim one row"); seen[r + x] = 1; }
        if ((k == 5) != (pair_of[r] >= 0)) return failg(1, "jg_gn_set_readings: kind 5 (correlated rectangular PMU) exactly on the rows jg_gn_create got as corr_row");
        t[d] = NoiseDev{r, k, k == 5 ? pair_of[r] : 0, (s1[d] ? 1 : 0) | (s2[d] ? 2 : 0), z1[d], v1[d], z2[d], k >= 2 ? v2[d] : 1.0};
    }
    for (int r = 0; r < h->m; ++r) if (!seen[r]) return failg(1, "jg_gn_set_readings: a row of the model belongs to no device");
    GN_HIP(hipStreamSynchronize(h->stream));
    hipFree(h->d_noise); h->d_noise = nullptr;
    GN_HIP(hipMalloc((void**)&h->d_noise, t.size() * sizeof(NoiseDev)));
    GN_HIP(jg::sync_copy(h->d_noise, t.data(), t.size() * sizeof(NoiseDev), hipMemcpyHostToDevice, h->stream));
    if (!h->d_noise_bad) GN_HIP(hipMalloc((void**)&h->d_noise_bad, sizeof(int)));
    h->n_noise = (int)ndev;
    return 0;
}

int jg_gn_draw_noise(jg_gn* h, uint64_t seed, double scale, int64_t first_realisation) {
    if (!h || !(scale >= 0.0) || first_realisation < 0) return failg(1, "jg_gn_draw_noise: bad argument");
    if (!h->d_noise) return failg(1, "jg_gn_draw_noise: call jg_gn_set_readings first");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipMemsetAsync(h->d_noise_bad, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_gn_noise, dim3((unsigned)((h->n_noise + 15) / 16), (unsigned)(h->ld / 64)), dim3(64, 16), 0, h->stream, h->d_noise, h->n_noise,
                       (unsigned long long)seed, scale, (long long)first_realisation, h->d_mean, h->d_w, h->m, h->ld, h->batch, h->d_noise_bad);
    GN_HIP(hipGetLastError());
    int bad = 0;
    GN_HIP(jg::sync_copy(&bad, h->d_noise_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (bad) return failg(1, "jg_gn_draw_noise: the variance of a measurement is zero or not finite (errorVariance): a squared-current / rectangular PMU reading of magnitude zero");
    return 0;
}

int jg_gn_get_measurement(jg_gn* h, double* mean, double* wdiag, double* woff) {
    if (!h || !mean || !wdiag) return failg(1, "jg_gn_get_measurement: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (int rc = get_rows(h, h->d_mean, mean, h->m)) return rc;
    if (int rc = get_rows(h, h->d_w, wdiag, h->m)) return rc;
    if (woff && h->ncorr > 0) return get_rows(h, h->d_w + (size_t)h->m * h->ld, woff, h->ncorr);
    return 0;
}

int jg_gn_set_voltage(jg_gn* h, const double* vm, const double* va, int64_t stride) {
    if (!h || !vm || !va) return failg(1, "jg_gn_set_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (int rc = put_rows(h, h->d_vm, vm, stride, h->n)) return rc;
    return put_rows(h, h->d_va, va, stride, h->n);
}

int jg_gn_snapshot_voltage(jg_gn* h) {
    if (!h) return failg(1, "jg_gn_snapshot_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    const size_t bytes = (size_t)h->n * h->ld * 8;
    if (!h->d_vm0) { GN_HIP(hipMalloc((void**)&h->d_vm0, bytes)); GN_HIP(hipMalloc((void**)&h->d_va0, bytes)); }
    GN_HIP(hipMemcpyAsync(h->d_vm0, h->d_vm, bytes, hipMemcpyDeviceToDevice, h->stream));
    GN_HIP(hipMemcpyAsync(h->d_va0, h->d_va, bytes, hipMemcpyDeviceToDevice, h->stream));
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_gn_restore_voltage(jg_gn* h) {
    if (!h || !h->d_vm0) return failg(1, "jg_gn_restore_voltage: no snapshot");
    if (int rc = set_device(h)) return rc;
    const size_t bytes = (size_t)h->n * h->ld * 8;
    GN_HIP(hipMemcpyAsync(h->d_vm, h->d_vm0, bytes, hipMemcpyDeviceToDevice, h->stream));
    GN_HIP(hipMemcpyAsync(h->d_va, h->d_va0, bytes, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int jg_gn_get_voltage(jg_gn* h, double* vm, double* va) {
    if (!h || !vm || !va) return failg(1, "jg_gn_get_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (int rc = get_rows(h, h->d_vm, vm, h->n)) return rc;
    return get_rows(h, h->d_va, va, h->n);
}

int jg_gn_increment(jg_gn* h, double* maxinc) {
    if (!h) return failg(1, "jg_gn_increment: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipMemsetAsync(h->eng.status, 0, (size_t)h->ld * 4, h->stream));
    {
        if (int rc = launch_increment(h, nullptr)) return rc;
    }
    launch_check(h, 0);
    GN_HIP(hipGetLastError());
    GN_HIP(hipStreamSynchronize(h->stream));
    std::vector<int> st(h->ld);
    GN_HIP(jg::sync_copy(st.data(), h->eng.status, (size_t)h->ld * 4, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b) if (st[b] & 4) return failg(3, "jg_gn_increment: zero or non-finite pivot (singular gain matrix)");
    if (maxinc) GN_HIP(jg::sync_copy(maxinc, h->d_maxinc, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_gn_solve(jg_gn* h) {
    if (!h) return failg(1, "jg_gn_solve: bad argument");
    if (int rc = set_device(h)) return rc;
    launch_update(h, nullptr);
    hipLaunchKernelGGL(k_gn_add_iter, dim3((h->ld + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->ld);
    GN_HIP(hipGetLastError());
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_gn_run(jg_gn* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status) {
    if (!h || max_iter < 0 || !(tol > 0.0)) return failg(1, "jg_gn_run: bad argument");
    if (int rc = set_device(h)) return rc;
    if (!h->exec) {
        std::lock_guard<std::mutex> lk(jg::capture_mutex());
        GN_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        hipMemsetAsync(h->d_counter, 0, sizeof(int), h->stream);
        // the first pass must see every group active: the flags of the PREVIOUS check drive group skipping
        int rc = launch_increment(h, h->d_group);
        hipMemsetAsync(h->d_group, 0, (size_t)(h->ld / 64) * sizeof(int), h->stream);
        launch_check(h, 1);
        launch_update(h, h->d_active);
        hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(int), hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
        if (rc) return rc;
        GN_HIP(e);
        GN_HIP(hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0));
    }
    const double params[2] = {tol, (double)max_iter};
    GN_HIP(hipMemcpyAsync(h->d_params, params, sizeof(params), hipMemcpyHostToDevice, h->stream));
    GN_HIP(hipMemsetAsync(h->d_iters, 0, (size_t)h->ld * 4, h->stream));          // acStateEstimation.jl:1298
    GN_HIP(hipMemsetAsync(h->eng.status, 0, (size_t)h->ld * 4, h->stream));
    GN_HIP(hipMemsetAsync(h->d_group, 0xff, (size_t)(h->ld / 64) * sizeof(int), h->stream));
    // (round 5, as jg_nr_run: while the handle's iterations are SHORT the host arms the pinned word and polls it instead of paying a stream synchronise per
    // iteration; a long iteration -- config 4 at 512 lanes: 4.4 ms -- blocks as before, see wait_verdict in jg_nr.hip.  JG_POLL=0 switches polling off)
    static const bool poll = jg::knob("POLL", 1) != 0;
    for (int64_t it = 0; it <= max_iter; ++it) {                                   // :1303
        const bool spin = poll && h->wait_us <= 800.0;
        if (spin) *(volatile int*)h->h_counter = -1;
        GN_HIP(hipGraphLaunch(h->exec, h->stream));
        const auto t0 = std::chrono::steady_clock::now();
        auto elapsed = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
        bool block = !spin;
        if (spin) {
            volatile int* w = (volatile int*)h->h_counter;
            for (long spins = 0; *w == -1; ++spins) {
                if ((spins & 63) == 63) {
                    const double us = elapsed();
                    if (us > 2.0e6) { block = true; break; }
                    if (us > 1600.0) std::this_thread::yield();
                }
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
        }
        if (block) GN_HIP(hipStreamSynchronize(h->stream));
        h->wait_us = 0.5 * h->wait_us + 0.5 * elapsed();
        if (*h->h_counter == 0) break;
    }
    GN_HIP(hipStreamSynchronize(h->stream));
    h->ran = true;
    if (iters) GN_HIP(jg::sync_copy(iters, h->d_iters, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (status) GN_HIP(jg::sync_copy(status, h->d_status, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

namespace {
// se.objective of every scenario from the residual the handle holds, into d_obj (stream-ordered, not synchronised)
int launch_objective(jg_gn* h) {
    if (!h->d_obj) {
        h->obj_chunks = (h->m + OBJ_ROWS - 1) / OBJ_ROWS;
        GN_HIP(hipMalloc((void**)&h->d_obj, (size_t)h->ld * 8));
        GN_HIP(hipMalloc((void**)&h->d_objpart, (size_t)h->obj_chunks * h->ld * 8));
        std::string err;
        if (jg::upload(&h->d_corr, h->corr_row, err, h->stream)) return failg(2, err);
    }
    hipLaunchKernelGGL(k_gn_obj_partial, dim3(h->obj_chunks, h->ld / 64), dim3(64, 16), 0, h->stream, h->d_res, h->d_w, h->d_objpart, h->m, h->ld, h->batch);
    hipLaunchKernelGGL(k_gn_obj_final, dim3(h->ld / 64), dim3(64), 0, h->stream, h->d_objpart, h->obj_chunks, h->d_res, h->d_w + (size_t)h->m * h->ld, h->d_corr, h->ncorr,
                       h->d_obj, h->ld, h->batch);
    GN_HIP(hipGetLastError());
    return 0;
}
int launch_pack(jg_gn* h, double* dst) {
    if (int rc = launch_objective(h)) return rc;
    const long long stride = 2LL * h->n + 3;
    hipLaunchKernelGGL(k_gn_pack_bus, dim3((h->n + 63) / 64, (h->ld + 63) / 64, 2), dim3(64, 8), 0, h->stream, h->d_vm, h->d_va, dst, h->n, h->ld, h->batch, stride);
    hipLaunchKernelGGL(k_gn_pack_tail, dim3((h->batch + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->d_status, h->d_obj, dst, h->batch, stride, 2 * h->n);
    GN_HIP(hipGetLastError());
    return 0;
}
}  // namespace

int jg_gn_get_objective(jg_gn* h, double* objective) {
    if (!h || !objective) return failg(1, "jg_gn_get_objective: bad argument");
    if (int rc = set_device(h)) return rc;
    if (int rc = launch_objective(h)) return rc;
    GN_HIP(jg::sync_copy(objective, h->d_obj, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_gn_pack_results_device(jg_gn* h, double* dst_dev) {
    if (!h || !dst_dev) return failg(1, "jg_gn_pack_results_device: bad argument");
    if (!h->ran) return failg(1, "jg_gn_pack_results_device: no stateEstimation! run on this handle yet (jg_gn_run): iterations and status are undefined");
    if (int rc = set_device(h)) return rc;
    if (int rc = launch_pack(h, dst_dev)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_gn_allgather_results(jg_gn* h, jg_comm* c, double* dst_dev) {
    if (!h || !c || !dst_dev) return failg(1, "jg_gn_allgather_results: bad argument");
    if (!h->ran) return failg(1, "jg_gn_allgather_results: no stateEstimation! run on this handle yet (jg_gn_run)");
    if (jg::comm_device(c) != h->device) return failg(1, "jg_gn_allgather_results: communicator and handle live on different devices");
    if (int rc = set_device(h)) return rc;
    const size_t count = (size_t)h->batch * (2 * (size_t)h->n + 3);
    double* mine = dst_dev + (size_t)jg::comm_rank(c) * count;    // in-place all-gather: this rank's record sits in its own block
    if (int rc = launch_pack(h, mine)) return rc;
    if (int rc = jg::comm_allgather(c, mine, dst_dev, count, h->stream)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_gn_get_maps(jg_gn* h, int8_t* type, int64_t* hcolptr, int64_t* hrowval) {
    if (!h) return failg(1, "jg_gn_get_maps: bad argument");
    if (type) std::memcpy(type, h->type.data(), h->type.size());
    if (hcolptr) std::memcpy(hcolptr, h->hcolptr.data(), h->hcolptr.size() * 8);
    if (hrowval) std::memcpy(hrowval, h->hrowval.data(), h->hrowval.size() * 8);
    return 0;
}

int jg_gn_get_jacobian(jg_gn* h, double* nzval) {
    if (!h || !nzval) return failg(1, "jg_gn_get_jacobian: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t((size_t)h->nslots * 2 * h->ld);
    GN_HIP(jg::sync_copy(t.data(), h->d_Hs, t.size() * 8, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (int64_t k = 0; k < h->nnzH; ++k) nzval[(size_t)b * h->nnzH + k] = t[((size_t)(h->hmap[k] >> 1) * h->ld + b) * 2 + (h->hmap[k] & 1)];     // hmap = slot * 2 + component
    return 0;
}

int jg_gn_get_residual(jg_gn* h, double* res) {
    if (!h || !res) return failg(1, "jg_gn_get_residual: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    return get_rows(h, h->d_res, res, h->m);
}

int jg_gn_get_increment(jg_gn* h, double* inc) {
    if (!h || !inc) return failg(1, "jg_gn_get_increment: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t((size_t)h->n * 2 * h->ld);                               // device [n][ld][2]
    GN_HIP(jg::sync_copy(t.data(), h->d_inc, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b)                                             // [theta_1..n, V_1..n] like se.increment
        for (int i = 0; i < h->n; ++i) {
            inc[(size_t)b * 2 * h->n + i] = t[((size_t)i * h->ld + b) * 2];
            inc[(size_t)b * 2 * h->n + h->n + i] = t[((size_t)i * h->ld + b) * 2 + 1];
        }
    return 0;
}

int jg_gn_get_iteration(jg_gn* h, int32_t* iters) {
    if (!h || !iters) return failg(1, "jg_gn_get_iteration: bad argument");
    if (int rc = set_device(h)) return rc;
    GN_HIP(hipStreamSynchronize(h->stream));
    GN_HIP(jg::sync_copy(iters, h->d_iters, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_gn_set_status(jg_gn* h, const int8_t* status, const int8_t* code) {
    if (!h || !status) return failg(1, "jg_gn_set_status: bad argument");
    for (int r = 0; r < h->m; ++r) if (status[r] != 0 && status[r] != 1) return failg(1, "jg_gn_set_status: status must be 0 or 1");
    if (code) {                                                 // magnitude <-> squared magnitude of a current: same two slots
        auto cls = [](int c) { return c == 4 ? 2 : (c == 5 ? 3 : c); };
        for (int r = 0; r < h->m; ++r)
            if (cls(code[r]) != cls(h->code[r])) return failg(1, "jg_gn_set_status: a type code may only switch between a current magnitude and its square");
        h->code.assign(code, code + h->m);
    }
    if (int rc = set_device(h)) return rc;
    for (int r = 0; r < h->m; ++r) {                            // se.type = status * code; the pattern keeps every row
        h->type[r] = (int8_t)(status[r] * h->code[r]);
        h->rows_host[r].type = h->type[r];
    }
    GN_HIP(hipStreamSynchronize(h->stream));
    GN_HIP(jg::sync_copy(h->d_rows, h->rows_host.data(), h->rows_host.size() * sizeof(RowDesc), hipMemcpyHostToDevice, h->stream));
    return 0;
}

int jg_gn_evaluate(jg_gn* h) {
    if (!h) return failg(1, "jg_gn_evaluate: bad argument");
    if (int rc = set_device(h)) return rc;
    launch_rows(h);
    GN_HIP(hipGetLastError());
    GN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_gn_residual_test(jg_gn* h, double* max_nres, int32_t* index) {
    if (!h || !max_nres || !index) return failg(1, "jg_gn_residual_test: bad argument");
    if (int rc = set_device(h)) return rc;
    constexpr int ROWS_PER = 256;
    if (!h->d_nres) {                                            // pair lists: every (slot, slot) of a row with its Z entry
        const jg::BlockSymbolic& S = h->eng.plan->S;
        std::vector<int> pp(h->m + 1, 0), pa, pb, pz;
        for (int r = 0; r < h->m; ++r) {
            const RowDesc& d = h->rows_host[r];
            for (int x = 0; x < d.nslots; ++x)
                for (int y = x; y < d.nslots; ++y) {
                    int sa = d.slot0 + x, sc = d.slot0 + y;
                    int ka = S.iperm[h->slot_bus_host[sa]], kc = S.iperm[h->slot_bus_host[sc]];
                    if (ka > kc) { std::swap(sa, sc); std::swap(ka, kc); }     // h_a Z_ac h_c' == h_c Z_ca h_a'
                    const int e = ka == kc ? S.diag[ka] : jg::entry_of(S, ka, kc);
                    if (e < 0 || (ka == kc && sa != sc)) return failg(1, "jg_gn_residual_test: a row's buses are not a clique of the gain pattern");
                    pa.push_back(sa); pb.push_back(sc); pz.push_back(e);
                }
            pp[r + 1] = (int)pa.size();
        }
        if (pa.empty()) { pa.push_back(0); pb.push_back(0); pz.push_back(0); }
        std::string err;
        h->amax_chunks = (h->m + ROWS_PER - 1) / ROWS_PER;
        if (jg::upload(&h->d_pair_ptr, pp, err, h->stream) || jg::upload(&h->d_pa, pa, err, h->stream) || jg::upload(&h->d_pb, pb, err, h->stream) ||
            jg::upload(&h->d_pz, pz, err, h->stream)) return failg(2, err);
        GN_HIP(hipMalloc((void**)&h->d_nres, (size_t)h->m * h->ld * 8));
        GN_HIP(hipMalloc((void**)&h->d_amax_v, (size_t)h->amax_chunks * h->ld * 8));
        GN_HIP(hipMalloc((void**)&h->d_amax_i, (size_t)h->amax_chunks * h->ld * 4));
        GN_HIP(hipMalloc((void**)&h->d_bad_v, (size_t)h->ld * 8));
        GN_HIP(hipMalloc((void**)&h->d_bad_i, (size_t)h->ld * 4));
    }
    GN_HIP(hipMemsetAsync(h->eng.status, 0, (size_t)h->ld * 4, h->stream));
    {
        launch_rows(h);                                              // residual and Jacobian at the CURRENT state
        launch_gain(h);
        const bool jordan = h->eng.jordan;
        h->eng.jordan = false;                                       // the selected inverse walks U itself: plain rows from the tasks
        const int rcf = h->eng.factor(h->stream, nullptr, h->d_rhs, jg::GroupSel{});
        h->eng.jordan = jordan;                                      // (the next increment factorises again before it solves)
        if (rcf) return failg(rcf, h->eng.error);
        if (int rc = h->eng.selected_inverse(h->stream, jg::GroupSel{})) return failg(rc, h->eng.error);
    }
    ProjArgs a{h->d_pair_ptr, h->d_pa, h->d_pb, h->d_pz, h->d_slot_bus, h->d_Hs, h->eng.Zs, h->d_res, h->d_w, h->d_nres, h->m, h->slack0, h->ld};
    hipLaunchKernelGGL(k_gn_project, dim3((h->m + GN_ROWS - 1) / GN_ROWS, h->ld / 64), dim3(64, 4), 0, h->stream, a);
    hipLaunchKernelGGL(k_gn_argmax, dim3(h->amax_chunks, h->ld / 64), dim3(64, 4), 0, h->stream, h->d_nres, h->d_amax_v, h->d_amax_i, h->m, ROWS_PER, h->ld);
    hipLaunchKernelGGL(k_gn_argmax2, dim3((h->ld + 255) / 256), dim3(256), 0, h->stream, h->d_amax_v, h->d_amax_i, h->d_bad_v, h->d_bad_i, h->amax_chunks, h->ld);
    GN_HIP(hipGetLastError());
    GN_HIP(hipStreamSynchronize(h->stream));
    std::vector<int> st(h->ld);
    GN_HIP(jg::sync_copy(st.data(), h->eng.status, (size_t)h->ld * 4, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b) if (st[b] & 4) return failg(3, "jg_gn_residual_test: zero or non-finite pivot (singular gain matrix)");
    GN_HIP(jg::sync_copy(max_nres, h->d_bad_v, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    GN_HIP(jg::sync_copy(index, h->d_bad_i, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_gn_get_normalized_residual(jg_gn* h, double* nres) {
    if (!h || !nres) return failg(1, "jg_gn_get_normalized_residual: bad argument");
    if (!h->d_nres) return failg(1, "jg_gn_get_normalized_residual: call jg_gn_residual_test first");
    if (int rc = set_device(h)) return rc;
    return get_rows(h, h->d_nres, nres, h->m);
}

int jg_gn_time_kernel(jg_gn* h, int kernel, int reps, double* mean_ms) {
    if (!h || reps < 1 || !mean_ms || kernel < 0 || kernel > 4) return failg(1, "jg_gn_time_kernel: bad argument");
    if (int rc = set_device(h)) return rc;
    hipEvent_t e0, e1;
    GN_HIP(hipEventCreate(&e0));
    GN_HIP(hipEventCreate(&e1));
    jg::StateUpdate none{};
    GN_HIP(hipStreamSynchronize(h->stream));
    GN_HIP(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) {
        if (kernel == 0) launch_rows(h);
        else if (kernel == 1) launch_gain(h);
        else if (kernel == 2) { if (int rc = h->eng.factor(h->stream, nullptr, h->d_rhs, jg::GroupSel{})) return failg(rc, h->eng.error); }
        else if (kernel == 3) { if (int rc = h->eng.backsolve(h->stream, h->d_inc, none, jg::GroupSel{})) return failg(rc, h->eng.error); }
        else { if (int rc = h->eng.selected_inverse(h->stream, jg::GroupSel{})) return failg(rc, h->eng.error); }    // (timing only: whatever the storage holds)
    }
    GN_HIP(hipEventRecord(e1, h->stream));
    GN_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GN_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *mean_ms = (double)ms / reps;
    return 0;
}

}  // extern "C"

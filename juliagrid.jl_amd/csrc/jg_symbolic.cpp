// jg_symbolic.cpp -- ordering, fill pattern, update-term lists and static schedules (host, integer only).
// See jg_symbolic.hpp for what this replaces in the reference.
#include "jg_symbolic.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace jg {

namespace {

// Exact minimum-degree elimination on the (small, sparse) bus graph.  Returns the elimination
// order and, for every eliminated vertex, its alive neighbourhood at elimination time
// (= the off-diagonal structure of that pivot row/column of the factor).
void min_degree(int n, const std::vector<std::vector<int>>& adj0, std::vector<int>& order,
                std::vector<std::vector<int>>& strct) {
    std::vector<std::vector<int>> adj = adj0;
    std::vector<char> done(n, 0);
    std::vector<int> mark(n, -1), deg(n), head(n + 1, -1), next(n, -1), prev(n, -1);
    auto insert = [&](int v) {
        int d = deg[v];
        next[v] = head[d]; prev[v] = -1;
        if (head[d] >= 0) prev[head[d]] = v;
        head[d] = v;
    };
    auto remove = [&](int v) {
        if (prev[v] >= 0) next[prev[v]] = next[v]; else head[deg[v]] = next[v];
        if (next[v] >= 0) prev[next[v]] = prev[v];
    };
    for (int i = n - 1; i >= 0; --i) { deg[i] = (int)adj[i].size(); insert(i); }
    order.resize(n);
    strct.assign(n, {});
    int mind = 0, stamp = 0;
    for (int k = 0; k < n; ++k) {
        while (head[mind] < 0) ++mind;
        int v = head[mind];
        remove(v);
        done[v] = 1;
        order[k] = v;
        std::vector<int> nb;
        for (int u : adj[v]) if (!done[u]) nb.push_back(u);
        for (int u : nb) {
            remove(u);
            ++stamp;
            std::vector<int>& au = adj[u];
            size_t m = 0;
            for (size_t s = 0; s < au.size(); ++s) { int w = au[s]; if (!done[w]) { mark[w] = stamp; au[m++] = w; } }
            au.resize(m);
            for (int w : nb) if (w != u && mark[w] != stamp) { au.push_back(w); mark[w] = stamp; }
            deg[u] = (int)au.size();
            insert(u);
            if (deg[u] < mind) mind = deg[u];
        }
        strct[k] = std::move(nb);
        std::vector<int>().swap(adj[v]);
    }
}

int find_in_row(const BlockSymbolic& S, int r, int c) {
    const int* b = S.e_col.data() + S.row_ptr[r];
    const int* e = S.e_col.data() + S.row_ptr[r + 1];
    const int* p = std::lower_bound(b, e, c);
    return (p != e && *p == c) ? (int)(p - S.e_col.data()) : -1;
}

// One launch per level.  Levels whose items carry long update lists (the dense tail of the
// elimination order) get several waves per item: the list is split across `wpi` waves and reduced
// through LDS, so the critical path of a level is ~ (terms / wpi) memory round trips instead of `terms`.
void schedule_by_level(const std::vector<int>& level, const std::vector<int>& work, int n_items, Schedule& sch, bool fuse_narrow_prefix = false) {
    int nlev = 0;
    for (int i = 0; i < n_items; ++i) nlev = std::max(nlev, level[i]);
    std::vector<int> cnt(nlev + 2, 0);
    for (int i = 0; i < n_items; ++i) cnt[level[i] + 1]++;
    for (int l = 0; l <= nlev; ++l) cnt[l + 1] += cnt[l];
    sch.items.assign(n_items, 0);
    {
        std::vector<int> pos(cnt.begin(), cnt.end() - 1);
        for (int i = 0; i < n_items; ++i) sch.items[pos[level[i]]++] = i;
    }
    sch.n_levels = nlev;
    sch.task_ptr.assign(1, 0);
    sch.step_ptr.assign(1, 0);
    sch.launches.clear();
    sch.step_wpi.clear();
    int first_level = 1;
    if (fuse_narrow_prefix) {
        // Leading levels that are narrow (a handful of items, short lists) are latency-bound: a launch per level
        // costs more than the work.  One workgroup of 16 waves per scenario group walks them as steps.
        constexpr int FW = 16;
        int last = 0;
        for (int l = 1; l <= nlev; ++l) {
            int items = cnt[l + 1] - cnt[l], tot = 0;
            for (int i = cnt[l]; i < cnt[l + 1]; ++i) tot += work[sch.items[i]];
            if (items == 0) continue;
            if (items > FW || tot > 64 * FW) break;
            last = l;
        }
        if (last >= 2) {
            Launch L;
            L.fused = 1; L.waves = FW; L.wpi = 1; L.chunk = 0;
            L.task_begin = (int)sch.task_ptr.size() - 1;
            L.item_begin = cnt[1]; L.item_end = cnt[last + 1];
            for (int l = 1; l <= last; ++l) {
                int b = cnt[l], e = cnt[l + 1];
                if (b == e) continue;
                std::stable_sort(sch.items.begin() + b, sch.items.begin() + e, [&](int x, int y) { return work[x] > work[y]; });
                int wpi = 1;
                while (wpi * 2 * (e - b) <= FW && work[sch.items[b]] > 2 * wpi) wpi *= 2;
                sch.step_ptr.push_back(e);
                sch.step_wpi.push_back(wpi);
            }
            sch.task_ptr.push_back((int)sch.step_ptr.size() - 1);
            L.task_end = (int)sch.task_ptr.size() - 1;
            sch.launches.push_back(L);
            first_level = last + 1;
        }
    }
    for (int l = first_level; l <= nlev; ++l) {
        int b = cnt[l], e = cnt[l + 1];
        if (b == e) continue;
        // heaviest items first inside the level (they start first on the device)
        std::stable_sort(sch.items.begin() + b, sch.items.begin() + e, [&](int x, int y) { return work[x] > work[y]; });
        const int maxw = work[sch.items[b]];
        Launch L;
        L.task_begin = (int)sch.task_ptr.size() - 1;
        int wpi = 1;
        while (wpi < 16 && maxw > 6 * wpi) wpi *= 2;
        if (e - b >= 2048) wpi = std::min(wpi, 2);          // wide levels already fill the chip
        L.wpi = wpi;
        L.waves = std::max(4, wpi);
        const int chunk = wpi == 1 ? 16 : L.waves / wpi;
        L.chunk = chunk; L.item_begin = b; L.item_end = e;
        for (int s = b; s < e; s += chunk) {
            sch.step_ptr.push_back(std::min(s + chunk, e));
            sch.step_wpi.push_back(wpi);
            sch.task_ptr.push_back((int)sch.step_ptr.size() - 1);
        }
        L.task_end = (int)sch.task_ptr.size() - 1;
        sch.launches.push_back(L);
    }
}

}  // namespace

int analyze(int n, const int* rowptr, const int* col, int policy, BlockSymbolic& S) {
    (void)policy;
    S = BlockSymbolic();
    S.n = n;
    if (n <= 0) return 1;
    // adjacency without the diagonal; verify structural symmetry and diagonal presence
    std::vector<std::vector<int>> adj(n);
    std::vector<char> has_diag(n, 0);
    for (int i = 0; i < n; ++i)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            int j = col[p];
            if (j < 0 || j >= n) return 1;
            if (j == i) has_diag[i] = 1; else adj[i].push_back(j);
        }
    for (int i = 0; i < n; ++i) {
        if (!has_diag[i]) return 1;
        std::sort(adj[i].begin(), adj[i].end());
        adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
    }
    for (int i = 0; i < n; ++i)
        for (int j : adj[i])
            if (!std::binary_search(adj[j].begin(), adj[j].end(), i)) return 1;

    std::vector<std::vector<int>> strct;
    min_degree(n, adj, S.perm, strct);
    S.iperm.assign(n, 0);
    for (int k = 0; k < n; ++k) S.iperm[S.perm[k]] = k;
    for (int k = 0; k < n; ++k) {
        for (int& u : strct[k]) u = S.iperm[u];
        std::sort(strct[k].begin(), strct[k].end());
    }

    // row patterns of L + D + U (pivot numbering)
    std::vector<int> lcount(n, 0);
    for (int k = 0; k < n; ++k) for (int i : strct[k]) lcount[i]++;
    S.row_ptr.assign(n + 1, 0);
    for (int r = 0; r < n; ++r) S.row_ptr[r + 1] = S.row_ptr[r] + lcount[r] + 1 + (int)strct[r].size();
    S.n_entries = S.row_ptr[n];
    S.e_col.assign(S.n_entries, 0);
    S.e_row.assign(S.n_entries, 0);
    {
        std::vector<int> fill(S.row_ptr.begin(), S.row_ptr.end() - 1);
        for (int k = 0; k < n; ++k) for (int i : strct[k]) S.e_col[fill[i]++] = k;     // ascending k
        for (int r = 0; r < n; ++r) {
            S.e_col[fill[r]++] = r;
            for (int j : strct[r]) S.e_col[fill[r]++] = j;
        }
    }
    S.diag.assign(n, -1);
    S.e_diag.assign(S.n_entries, -1);
    for (int r = 0; r < n; ++r)
        for (int e = S.row_ptr[r]; e < S.row_ptr[r + 1]; ++e) {
            S.e_row[e] = r;
            if (S.e_col[e] == r) S.diag[r] = e;
        }
    for (int e = 0; e < S.n_entries; ++e)
        if (S.e_row[e] > S.e_col[e]) S.e_diag[e] = S.diag[S.e_col[e]];

    // source positions in the caller's block CSR
    S.e_src.assign(S.n_entries, -1);
    for (int i = 0; i < n; ++i)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            int e = find_in_row(S, S.iperm[i], S.iperm[col[p]]);
            if (e < 0) return 1;
            if (S.e_src[e] >= 0) return 1;                    // duplicate block in the input pattern
            S.e_src[e] = p;
        }

    // update terms: pivot k contributes -L(i,k) U(k,j) to every (i,j) in struct(k)^2
    S.t_ptr.assign(S.n_entries + 1, 0);
    for (int k = 0; k < n; ++k) {
        const std::vector<int>& s = strct[k];
        for (int i : s) for (int j : s) S.t_ptr[find_in_row(S, i, j) + 1]++;
    }
    for (int e = 0; e < S.n_entries; ++e) S.t_ptr[e + 1] += S.t_ptr[e];
    S.n_terms = S.t_ptr[S.n_entries];
    S.t_a.assign((size_t)S.n_terms, 0);
    S.t_d.assign((size_t)S.n_terms, 0);
    S.t_b.assign((size_t)S.n_terms, 0);
    {
        std::vector<int> fill(S.t_ptr.begin(), S.t_ptr.end() - 1);
        std::vector<int> lid, uid;
        for (int k = 0; k < n; ++k) {
            const std::vector<int>& s = strct[k];
            lid.resize(s.size()); uid.resize(s.size());
            for (size_t a = 0; a < s.size(); ++a) { lid[a] = find_in_row(S, s[a], k); uid[a] = find_in_row(S, k, s[a]); }
            for (size_t a = 0; a < s.size(); ++a)
                for (size_t b = 0; b < s.size(); ++b) {
                    int t = find_in_row(S, s[a], s[b]);
                    S.t_a[fill[t]] = lid[a];
                    S.t_d[fill[t]] = S.diag[k];
                    S.t_b[fill[t]] = uid[b];
                    fill[t]++;
                }
        }
    }

    // entry dependency levels (pivots ascending: all sources of pivot-k entries have smaller pivots);
    // D(k), U(k,.) and Lh(.,k) have no mutual dependency, so one level per pivot "generation"
    S.e_level.assign(S.n_entries, 0);
    {
        std::vector<int> acc(S.n_entries, 0);
        for (int k = 0; k < n; ++k) {
            const std::vector<int>& s = strct[k];
            const int d = S.diag[k];
            S.e_level[d] = acc[d] + 1;
            for (int j : s) {
                const int u = find_in_row(S, k, j), l = find_in_row(S, j, k);
                S.e_level[u] = acc[u] + 1;
                S.e_level[l] = acc[l] + 1;
            }
            for (int i : s) {
                const int li = std::max(S.e_level[find_in_row(S, i, k)], S.e_level[d]);
                for (int j : s) {
                    const int t = find_in_row(S, i, j);
                    acc[t] = std::max(acc[t], std::max(li, S.e_level[find_in_row(S, k, j)]));
                }
            }
        }
    }

    // triangular-solve row lists and levels
    S.l_ptr.assign(n + 1, 0);
    S.u_ptr.assign(n + 1, 0);
    for (int r = 0; r < n; ++r) { S.l_ptr[r + 1] = S.l_ptr[r] + lcount[r]; S.u_ptr[r + 1] = S.u_ptr[r] + (int)strct[r].size(); }
    S.l_ent.assign(S.l_ptr[n], 0); S.l_col.assign(S.l_ptr[n], 0);
    S.u_ent.assign(S.u_ptr[n], 0); S.u_col.assign(S.u_ptr[n], 0);
    for (int r = 0; r < n; ++r) {
        int lp = S.l_ptr[r], up = S.u_ptr[r];
        for (int e = S.row_ptr[r]; e < S.row_ptr[r + 1]; ++e) {
            int c = S.e_col[e];
            if (c < r) { S.l_ent[lp] = e; S.l_col[lp++] = c; }
            else if (c > r) { S.u_ent[up] = e; S.u_col[up++] = c; }
        }
    }
    S.y_level.assign(n, 1);
    S.bwd_level.assign(n, 1);
    for (int r = 0; r < n; ++r)
        for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) {
            const int c = S.l_col[p];
            S.y_level[r] = std::max(S.y_level[r], 1 + std::max(S.y_level[c], std::max(S.e_level[S.l_ent[p]], S.e_level[S.diag[c]])));
        }
    for (int r = n - 1; r >= 0; --r)
        for (int p = S.u_ptr[r]; p < S.u_ptr[r + 1]; ++p) S.bwd_level[r] = std::max(S.bwd_level[r], S.bwd_level[S.u_col[p]] + 1);

    {
        std::vector<int> level(S.n_entries + n), work(S.n_entries + n), uw(n);
        for (int e = 0; e < S.n_entries; ++e) { level[e] = S.e_level[e]; work[e] = S.t_ptr[e + 1] - S.t_ptr[e]; }
        for (int r = 0; r < n; ++r) {
            level[S.n_entries + r] = S.y_level[r];
            work[S.n_entries + r] = S.l_ptr[r + 1] - S.l_ptr[r];
            uw[r] = S.u_ptr[r + 1] - S.u_ptr[r];
        }
        schedule_by_level(level, work, S.n_entries + n, S.fact);
        schedule_by_level(S.bwd_level, uw, n, S.bwd, /*fuse_narrow_prefix=*/true);
    }
    return 0;
}

}  // namespace jg

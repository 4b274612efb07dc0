// jg_symbolic.cpp -- ordering, fill pattern, update-term lists and static schedules (host, integer only).
// See jg_symbolic.hpp for what this replaces in the reference.
#include "jg_symbolic.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <type_traits>
#include <iterator>
#include <queue>
#include <system_error>
#include <thread>
#include <cstdio>
#include <cstring>
#include <string>
#include <cstdlib>
#include <numeric>

namespace jg {

int knob(const char* name, int unset) {
    static const char* const known[] = {"TRACE", "PLAN_CACHE", "POLL", "PLAN_THREADS", "PLAN_TIMING", "HOST_TIMING",
                                        "TOP_PW", "TOP_FUSE", "TOP_SYM", "JORDAN", "CHAIN_SMALL", "NO_PREFACTOR", "LANES_INPLACE", "TOP_LEVEL", "ROW_TASKS", "ORDER_CHECK", "TOP_PROFILE", "SINGLE"};
    bool ok = false;
    for (const char* k : known) ok = ok || std::strcmp(k, name) == 0;
    if (!ok) return unset;
    const std::string var = std::string("JG_") + name;
    const char* e = getenv(var.c_str());
    return e ? atoi(e) : unset;
}


void build_tables(BlockSymbolic& S);
constexpr int NO_TOP_LEVEL = 0x7fffffff;
void build_top(BlockSymbolic& S, int top_level, int soft_cap, int struct_min, int mid_mmin, int mid_strict);

namespace {

// Elimination order of the (small, sparse) bus graph: greedy MINIMUM LOCAL FILL with a height penalty.
//   score(v) = 20 * fill(v) + h(v)^2       fill(v) = pairs of alive neighbours of v that are not adjacent yet,
//                                          h(v)    = longest chain of already eliminated vertices hanging below v
// (ties: smaller degree, then smaller index => deterministic).  Against exact minimum degree this gives ~15-20 % fewer
// update terms AND 40 % fewer dependency levels on transmission grids (ACTIVSg10k: 293k -> 248k terms, elimination
// tree height 152 -> 84; 9241-bus PEGASE-shaped grid: 87k -> 80k, 97 -> 58): the fill term does what minimum degree
// approximates, the height term stops the greedy choice from growing one long chain when an equally cheap vertex in a
// shallow part of the graph is available (quadratic: free near the leaves, decisive near the top; a linear penalty
// 4 fill + 3 h gave 237k terms / 101 levels).  Both numbers are what the device pays for: terms are HBM traffic, levels are
// dependent launches.  Returns the order and, for every eliminated vertex, its alive neighbourhood at elimination time
// (= the off-diagonal structure of that pivot row/column of the factor).  Cost: O(sum over eliminations of
// |two-hop neighbourhood| * degree^2) -- 0.1 s for 10 000 buses.
bool elimination_order(int n, std::vector<std::vector<int>>&& adj0, std::vector<int>& order,
                       std::vector<std::vector<int>>& strct) {
    bool exact = true;
    std::vector<std::vector<int>> adj = std::move(adj0);       // sorted, alive neighbours only
    std::vector<char> done(n, 0);
    std::vector<int> hv(n, 0), mark(n, -1), seen(n, -1), hits(n, 0);
    std::vector<long long> cur(n, 0), fillv(n, 0);             // fillv: the CURRENT fill of every alive vertex (exact, kept up to date)
    int stamp = 0;
    auto fill_of = [&](int v) -> long long {
        const std::vector<int>& nb = adj[v];
        const long long d = (long long)nb.size();
        ++stamp;
        for (int a : nb) mark[a] = stamp;
        long long present = 0;                                 // adjacent pairs inside the neighbourhood, counted twice
        for (int a : nb) {                                     // (branch-free: the outcome of this test is a coin toss)
            const std::vector<int>& aa = adj[a];
            long long c = 0;
            for (size_t x = 0; x < aa.size(); ++x) c += mark[aa[x]] == stamp;
            present += c;
        }
        return d * (d - 1) / 2 - present / 2;
    };
    const long long wf = 20, wh = 0, wd = 0, wq = 1;           // score = 20 fill + height^2 (other weightings were measured in rounds 2-3: DESIGN_LOG)
    auto key_of = [&](int v) -> long long {                    // score, then degree, packed (degree < 2^20); fillv[v] must be current
        const long long d = (long long)adj[v].size(), h = hv[v];
        return ((wf * fillv[v] + wh * h + wd * d + wq * h * h) << 20) | std::min<long long>(d, (1 << 20) - 1);
    };
    auto key = [&](int v) -> long long { fillv[v] = fill_of(v); return key_of(v); };
    // The queue: an indexed 4-ary heap over the alive vertices ordered by (score key, index) -- a vertex whose score changes moves in place.
    // (The first build pushed a fresh entry per change and skipped the stale ones when they surfaced: 34 000 of the 44 000 pops of the 10 000-bus
    // grid.  Same minimum at every step, so the same order bit for bit.)
    std::vector<int> hp(n), hpos(n);
    int hn = 0;
    auto before = [&](int a, int b) { return cur[a] < cur[b] || (cur[a] == cur[b] && a < b); };
    auto sift_up = [&](int i) {
        const int v = hp[i];
        while (i > 0) { const int p = (i - 1) >> 2; if (!before(v, hp[p])) break; hp[i] = hp[p]; hpos[hp[i]] = i; i = p; }
        hp[i] = v; hpos[v] = i;
    };
    auto sift_down = [&](int i) {
        const int v = hp[i];
        for (;;) {
            const int c0 = 4 * i + 1;
            if (c0 >= hn) break;
            int best = c0;
            const int c1 = std::min(c0 + 4, hn);
            for (int c = c0 + 1; c < c1; ++c) if (before(hp[c], hp[best])) best = c;
            if (!before(hp[best], v)) break;
            hp[i] = hp[best]; hpos[hp[i]] = i; i = best;
        }
        hp[i] = v; hpos[v] = i;
    };
    auto rekey = [&](int v, long long k) {                      // v is alive: its key becomes k
        const long long old = cur[v];
        cur[v] = k;
        if (k < old) sift_up(hpos[v]); else if (k > old) sift_down(hpos[v]);
    };
    for (int v = 0; v < n; ++v) { cur[v] = key(v); hp[hn] = v; hpos[v] = hn; ++hn; }
    for (int i = (hn - 2) / 4; i >= 0 && hn > 1; --i) sift_down(i);
    order.clear(); order.reserve(n);
    strct.assign(n, {});
    const bool check = knob_set("ORDER_CHECK");     // tests: every incremental fill against a recount
    std::vector<int> merged, touched;
    std::vector<char> cadj;                                    // old adjacency inside the clique of the vertex being eliminated
    std::vector<long long> xcount, xr, oldpairs;
    std::vector<std::pair<int, int>> newedge;
    int epoch = 0;
    while (hn > 0) {
        const int v = hp[0];
        --hn;
        if (hn > 0) { hp[0] = hp[hn]; hpos[hp[0]] = 0; sift_down(0); }
        done[v] = 1;
        const int k = (int)order.size();
        order.push_back(v);
        std::vector<int> nb = std::move(adj[v]);
        std::vector<int>().swap(adj[v]);
        if (fillv[v] == 0) {
            // the neighbourhood of v is a clique already: no edge is created, so nobody outside it sees a change, and inside it a
            // vertex a only loses v -- the pairs (v, w) with w a neighbour of a outside the clique were its missing edges:
            // fill(a) -= deg(a) - |clique|  (deg(a) counted with v).  Exact, O(1) per neighbour: leaves and chains, most eliminations.
            const long long c = (long long)nb.size();
            for (int a : nb) {
                std::vector<int>& aa = adj[a];
                fillv[a] -= (long long)aa.size() - c;
                aa.erase(std::lower_bound(aa.begin(), aa.end(), v));
                hv[a] = std::max(hv[a], hv[v] + 1);
                rekey(a, key_of(a));
            }
            strct[k] = std::move(nb);
            continue;
        }
        // the edges the elimination creates: pairs of neighbours of v that are not adjacent yet (fillv[v] of them)
        newedge.clear();
        ++epoch;
        for (int a : nb) seen[a] = epoch;                      // seen == epoch: member of the new clique
        // ... and, in the same pass over the old neighbourhoods, what the new fill of every clique member a needs (exact, no recount).  With
        // Cold = the members a was adjacent to already, X = the others (its NEW neighbours), R = its old neighbours outside the clique:
        //   adjacent pairs among N'(a) = (N(a) \ {v}) u X  =  [pairs among N(a)] - |Cold| (those with v) + new edges inside Cold
        //                                                     + |X| |Cold| + C(|X|, 2) (the clique is complete) + sum_{x in X} |N(x) n R|
        // and pairs among N(a) = C(deg, 2) - fill(a), which is kept current.  Only the last sum walks lists -- those of the NEW neighbours of a,
        // not of all of them (the first build recounted fill(a) from scratch: 40 % of the ordering time; same order bit for bit).
        const int s = (int)nb.size();
        cadj.assign((size_t)s * s, 0); xcount.assign(s, 0); xr.assign(s, 0); oldpairs.assign(s, 0);
        for (int ia = 0; ia < s; ++ia) {
            const int a = nb[ia];
            ++stamp;
            for (int w : adj[a]) mark[w] = stamp;
            const long long d = (long long)adj[a].size();
            oldpairs[ia] = d * (d - 1) / 2 - fillv[a];
            for (int ix = 0; ix < s; ++ix) {
                const int x = nb[ix];
                if (x == a) continue;
                if (mark[x] == stamp) { cadj[(size_t)ia * s + ix] = 1; continue; }
                if (x > a) newedge.push_back(std::make_pair(a, x));
                xcount[ia]++;
                long long c = 0;                               // neighbours of x among the old neighbours of a outside the clique (v is in both lists)
                for (int w : adj[x]) c += (int)(mark[w] == stamp) & (int)(seen[w] != epoch) & (int)(w != v);
                xr[ia] += c;
            }
        }
        for (int a : nb) {                                     // v leaves, its neighbourhood becomes a clique
            std::vector<int>& aa = adj[a];
            merged.clear();                                    // (aa u nb) \ {v, a}: both sorted
            {
                size_t x = 0, y = 0;
                const size_t nx = aa.size(), ny = nb.size();
                while (x < nx || y < ny) {
                    int w;
                    if (y >= ny || (x < nx && aa[x] < nb[y])) w = aa[x++];
                    else { if (x < nx && aa[x] == nb[y]) ++x; w = nb[y++]; }
                    if (w != v && w != a) merged.push_back(w);
                }
            }
            aa.swap(merged);
            hv[a] = std::max(hv[a], hv[v] + 1);
        }
        // Whose score changed.  A vertex OUTSIDE the new clique keeps its neighbourhood; its fill counts the pairs of its neighbours that are not
        // adjacent, and the only pairs that became adjacent are the new edges: fill(u) -= 1 for every new edge (a, x) with u a common neighbour of
        // a and x -- exact, and a few list intersections per elimination (the greedy choice keeps fill(v) small) where the first build recomputed
        // the fill of every vertex with two neighbours in the clique from scratch (round 4: 19 -> 9 ms of the analysis of the 10 000-bus grid,
        // same order bit for bit).  The members of the clique (degree, height, neighbourhood all changed) are recomputed.
        touched.clear();
        for (const std::pair<int, int>& e : newedge) {
            ++stamp;
            for (int w : adj[e.first]) mark[w] = stamp;
            for (int u : adj[e.second])
                if (mark[u] == stamp && seen[u] != epoch) {
                    fillv[u] -= 1;
                    if (hits[u] != -epoch) { hits[u] = -epoch; touched.push_back(u); }
                }
        }
        for (int u : touched) rekey(u, key_of(u));
        for (int ia = 0; ia < s; ++ia) {
            long long cold = 0, newin = 0;
            const char* row = cadj.data() + (size_t)ia * s;
            for (int ix = 0; ix < s; ++ix) cold += row[ix];
            for (int ip = 0; ip < s; ++ip)                     // new edges inside Cold: pairs of old neighbours of a that were not adjacent
                if (row[ip]) for (int iq = ip + 1; iq < s; ++iq) newin += row[iq] & !cadj[(size_t)ip * s + iq];
            const long long x = xcount[ia];
            const long long present = oldpairs[ia] - cold + newin + x * cold + x * (x - 1) / 2 + xr[ia];
            const int a = nb[ia];
            const long long d = (long long)adj[a].size();
            fillv[a] = d * (d - 1) / 2 - present;
            if (check && fillv[a] != fill_of(a)) { fprintf(stderr, "jg_symbolic: incremental fill of vertex %d is %lld, recount %lld\n", a, fillv[a], fill_of(a)); exact = false; }
            rekey(a, key_of(a));
        }
        strct[k] = std::move(nb);
    }
    return exact;
}

// Postorder of the elimination tree (any topological order of the tree gives the same fill and the same levels): the
// pivots of a subtree become consecutive, and a child that forms a supernode with its parent (nested structure) is
// visited LAST, so parent = child + 1 -- that is what the backward-chain detection in build_tables looks for.  The
// greedy order interleaves subtrees and hides most chains.
void postorder(std::vector<int>& order, std::vector<std::vector<int>>& strct) {
    const int n = (int)order.size();
    std::vector<int> pos(n), parent(n, -1);
    int maxv = 0;
    for (int v : order) maxv = std::max(maxv, v);
    std::vector<int> where(maxv + 1, -1);
    for (int k = 0; k < n; ++k) where[order[k]] = k;
    for (int k = 0; k < n; ++k)
        for (int u : strct[k]) { const int p = where[u]; if (parent[k] < 0 || p < parent[k]) parent[k] = p; }
    std::vector<std::vector<int>> kids(n);
    for (int k = 0; k < n; ++k) if (parent[k] >= 0) kids[parent[k]].push_back(k);
    std::vector<int> size(n, 1);
    for (int k = 0; k < n; ++k) if (parent[k] >= 0) size[parent[k]] += size[k];          // children precede parents in `order`
    for (int p = 0; p < n; ++p) {
        auto nested = [&](int c) { return strct[c].size() == strct[p].size() + 1; };   // struct(c) = {p} + struct(p)
        std::stable_sort(kids[p].begin(), kids[p].end(), [&](int a, int b) {
            if (nested(a) != nested(b)) return nested(b);                                // the supernode child last
            return size[a] < size[b];                                                    // else the largest subtree last
        });
    }
    std::vector<int> seq;
    seq.reserve(n);
    std::vector<std::pair<int, size_t>> stack;
    for (int r = 0; r < n; ++r) {
        if (parent[r] >= 0) continue;
        stack.push_back({r, 0});
        while (!stack.empty()) {
            auto& top = stack.back();
            if (top.second < kids[top.first].size()) { const int c = kids[top.first][top.second++]; stack.push_back({c, 0}); }
            else { seq.push_back(top.first); stack.pop_back(); }
        }
    }
    std::vector<int> order2(n);
    std::vector<std::vector<int>> strct2(n);
    for (int i = 0; i < n; ++i) { order2[i] = order[seq[i]]; strct2[i] = std::move(strct[seq[i]]); }
    order.swap(order2);
    strct.swap(strct2);
}

int find_in_row(const BlockSymbolic& S, int r, int c) {
    const int* b = S.e_col.data() + S.row_ptr[r];
    const int* e = S.e_col.data() + S.row_ptr[r + 1];
    const int* p = std::lower_bound(b, e, c);
    return (p != e && *p == c) ? (int)(p - S.e_col.data()) : -1;
}

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// A helper thread where the host can give one (std::thread throws when it cannot: the C ABI above must never see that) -- else the work runs right here.
template <class F>
std::thread spawn_or_run(F&& f) {
    try { return std::thread(f); } catch (const std::system_error&) { f(); return std::thread(); }
}

// Levels -> segments -> chunks -> wave records.  `level[i]` (1-based; 0 = not scheduled) and `work[i]` (terms) per item;
// `emit(item, first_term_slot q0, stride, out)` fills one wave's records.
struct NoExtra { void operator()(int, std::vector<Segment>&, std::vector<Rec>&) const {} };

// `extra(level, segs, recs)` may append further segments of that level (backward chains); extra_levels = highest level it uses
template <class Fill, class Extra = NoExtra>
void build_replay(const std::vector<int>& level, const std::vector<int>& work, int T, std::vector<Segment>& segs,
                  std::vector<Rec>& recs, int& n_levels, Fill fill, Extra extra = Extra(), int extra_levels = 0, int max_wpi = 16,
                  const std::vector<long long>* locality = nullptr) {
    const int n_items = (int)level.size();
    int nlev = extra_levels;
    for (int i = 0; i < n_items; ++i) nlev = std::max(nlev, level[i]);
    std::vector<std::vector<int>> by(nlev + 1);
    for (int i = 0; i < n_items; ++i) if (level[i] > 0) by[level[i]].push_back(i);
    segs.clear(); recs.clear();
    {                                                             // the records of the whole table in one allocation (a lower bound is enough:
        size_t est = 0;                                           // one record per T terms and per item, idle waves of the last chunks on top)
        for (int i = 0; i < n_items; ++i) if (level[i] > 0) est += (size_t)std::max(1, (work[i] + T - 1) / T);
        recs.reserve(est + est / 4 + 1024);
    }
    n_levels = 0;
    constexpr int wide_cap = 4;
    constexpr bool has_extra = !std::is_same<Extra, NoExtra>::value;
    int nthreads = 1;
    if (n_items >= 4096) {
        const int hw = (int)std::max(1u, std::min(4u, std::thread::hardware_concurrency()));
        if (!has_extra && nlev >= 8) nthreads = hw;
    }
    // one level: its segments and records into `lsegs` / `lrecs` (rec_base relative to lrecs)
    auto do_level = [&](int l, std::vector<Segment>& lsegs, std::vector<Rec>& lrecs) {
        std::vector<int>& it = by[l];
        const bool wide = it.size() >= 2048;                      // wide levels already fill the chip: do not split short lists
        // ... and at most 4 waves share a long list there (the level that brings the bottom's terms to the top-owned entries holds
        // thousands of 20- to 70-term items: 26 112 workgroups at 8 waves per item, 139 us; 13 056 at 4, 117 us; at 2: 138 us)
        auto wpi_of = [&](int i) {
            int cap = wide && work[i] <= 4 * T ? 2 : max_wpi;
            if (wide && wide_cap > 0) cap = std::min(cap, wide_cap);
            return std::min(cap, pow2ceil(std::max(1, (work[i] + T - 1) / T)));
        };
        if (locality) {                                           // segments by waves per item (widest first); inside a segment the items
                                                                  // that share operands sit next to each other (same workgroup, same moment)
            struct Key { int wpi; long long loc; int item; };     // (keys once per item, not twice per comparison)
            std::vector<Key> keys(it.size());
            for (size_t x = 0; x < it.size(); ++x) keys[x] = Key{wpi_of(it[x]), (*locality)[it[x]], it[x]};
            std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.wpi != b.wpi ? a.wpi > b.wpi : a.loc < b.loc; });
            for (size_t x = 0; x < it.size(); ++x) it[x] = keys[x].item;
        } else
            std::stable_sort(it.begin(), it.end(), [&](int x, int y) { return work[x] > work[y]; });   // heaviest first
        size_t p = 0;
        while (p < it.size()) {
            const int wpi = wpi_of(it[p]);
            size_t q = p;
            while (q < it.size() && wpi_of(it[q]) == wpi) ++q;
            Segment sg{};
            sg.rec_base = (int)lrecs.size(); sg.wpi = wpi; sg.level = l; sg.items = (int)(q - p);
            const int slots = 16 / wpi;
            sg.nchunks = (sg.items + slots - 1) / slots;
            int wmax = 0;
            for (size_t x = p; x < q; ++x) wmax = std::max(wmax, work[it[x]]);
            sg.rpw = std::max(1, ((wmax + wpi - 1) / wpi + T - 1) / T);
            sg.last = 0;
            lrecs.resize(lrecs.size() + (size_t)sg.nchunks * 16 * sg.rpw);
            auto chunks = [&](int c0, int c1) {
                for (int c = c0; c < c1; ++c)
                    for (int w = 0; w < 16; ++w) {
                        const size_t idx = p + (size_t)c * slots + w / wpi;
                        Rec* r = &lrecs[sg.rec_base + ((size_t)c * 16 + w) * sg.rpw];
                        for (int j = 0; j < sg.rpw; ++j) { for (int k = 0; k < 16; ++k) r[j].w[k] = 0; r[j].w[0] = -1; }
                        if (idx < q) fill(it[idx], w % wpi, wpi, sg.rpw, r);
                    }
            };
            chunks(0, sg.nchunks);
            lsegs.push_back(sg);
            p = q;
        }
    };
    const bool rtiming = knob_set("PLAN_TIMING");
    auto rnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double r0 = rnow();
    if (nthreads > 1) {
        // levels are independent of each other (round 4: the analysis is what a cold power flow waits for): a few threads take them in turn,
        // the tables are stitched together in level order -- the result is the sequential one bit for bit
        std::vector<std::vector<Segment>> lsegs(nlev + 1);
        std::vector<std::vector<Rec>> lrecs(nlev + 1);
        std::atomic<int> next{1};
        std::vector<std::thread> pool;
        auto take = [&] { for (int l = next.fetch_add(1); l <= nlev; l = next.fetch_add(1)) do_level(l, lsegs[l], lrecs[l]); };
        for (int t = 1; t < nthreads; ++t) pool.push_back(spawn_or_run(take));
        take();
        for (std::thread& t : pool) if (t.joinable()) t.join();
        const double r1 = rnow();
        size_t total = 0;
        for (int l = 1; l <= nlev; ++l) total += lrecs[l].size();
        recs.reserve(total);
        for (int l = 1; l <= nlev; ++l) {
            const int base = (int)recs.size();
            for (Segment sg : lsegs[l]) { sg.rec_base += base; segs.push_back(sg); }
            recs.insert(recs.end(), lrecs[l].begin(), lrecs[l].end());
            if (!lsegs[l].empty()) { segs.back().last = 1; ++n_levels; }
        }
        if (rtiming) fprintf(stderr, "[jg plan]     replay of %d items, %d levels: levels %.1f ms on %d threads, stitched %.1f ms\n", n_items, nlev, r1 - r0, nthreads, rnow() - r1);
        return;
    }
    for (int l = 1; l <= nlev; ++l) {
        const size_t seg0 = segs.size();
        std::vector<Segment> lsegs;
        const size_t base = recs.size();
        {
            std::vector<Rec> lrecs;                               // (a level's records straight behind the table: one copy, as before)
            do_level(l, lsegs, lrecs);
            recs.insert(recs.end(), lrecs.begin(), lrecs.end());
        }
        for (Segment sg : lsegs) { sg.rec_base += (int)base; segs.push_back(sg); }
        extra(l, segs, recs);
        if (segs.size() > seg0) { segs.back().last = 1; ++n_levels; }
    }
}

}  // namespace

// Top tasks (jg_symbolic.hpp): choose the pivots, cut them into tasks, level the task tree, emit headers + data.
// A task is a CONNECTED piece of the elimination tree with one root: the root, then -- while the front has room -- pivots whose
// parent is already in the task, largest pivot number first (the supernode child of a pivot is its largest child, so a chain is
// followed to its end before siblings are taken).  Its front holds the task's pivots in ascending order, then struct(root);
// the elimination is dense over that front (a block outside a pivot's structure is an exact zero and stays one).
// mid_mmin > 0: GROUPED tasks (jg_symbolic.hpp) -- a task's front is capped by the geometry it will run with (15 rows at 16 scenarios per
// workgroup, 31 at 4, soft_cap at one), the smallest one that still leaves room for mid_mmin pivots (or for every task pivot of the root's
// subtree if there are fewer); mid_strict: pivots that would fit a smaller geometry are not absorbed (they form tasks of their own).
void build_top(BlockSymbolic& S, int top_level, int soft_cap, int struct_min, int mid_mmin, int mid_strict) {
    const int n = S.n;
    S.top_level = 0; S.top_task.clear(); S.top_data.clear(); S.top_launch.clear(); S.top_stack = 0; S.top_terms = 0; S.top_wgmap.clear();
    S.top_task_of.assign(n, -1);
    const bool want_jordan = S.jordan != 0;                      // granted below if the plan has tasks and none of them is grouped
    S.jordan = 0; S.n_jordan = 0;
    if (top_level <= 0 || (top_level == NO_TOP_LEVEL && struct_min <= 0)) return;   // NO_TOP_LEVEL: no pivot goes to a task for its level
    auto ssize = [&](int k) { return S.u_ptr[k + 1] - S.u_ptr[k]; };
    auto parent = [&](int k) { return ssize(k) ? S.u_col[S.u_ptr[k]] : -1; };
    std::vector<char> top(n, 0);
    int ntop = 0;
    if (top_level != NO_TOP_LEVEL) for (int k = 0; k < n; ++k) if (S.e_level[S.diag[k]] >= top_level) top[k] = 1;
    // fronts of struct_min blocks and more go to the tasks wherever they sit in the tree, and so do their ancestors (a task hands
    // its update matrix to the task of its parent): per (pivot, scenario) a task step costs the same whatever the front, a
    // level item costs per update term, and the two cross near 13 blocks (DESIGN_LOG.md 3.3)
    if (struct_min > 0) {
        for (int k = 0; k < n; ++k) if (ssize(k) >= struct_min) top[k] = 1;
        for (int k = 0; k < n; ++k) if (top[k] && parent(k) >= 0) top[parent(k)] = 1;
    }
    for (int k = 0; k < n; ++k) if (top[k]) { ++ntop; if (ssize(k) + 1 > TOP_FRONT_MAX) return; }
    if (ntop == 0) return;
    struct Task { std::vector<int> piv; int m, e, parent, level, cls, stack, lg; std::vector<int> kids; int root() const { return piv.back(); } };
    std::vector<Task> tasks;
    std::vector<int> sub(n, 1);                                  // subtree sizes: the descendants of k are k - sub[k] + 1 .. k - 1 (postorder)
    for (int k = 0; k < n; ++k) if (parent(k) >= 0) sub[parent(k)] += sub[k];
    std::vector<int> tsub(n, 0);                                 // task pivots in the subtree of k
    for (int k = 0; k < n; ++k) { if (top[k]) tsub[k]++; if (parent(k) >= 0) tsub[parent(k)] += tsub[k]; }
    const int gcap[3] = {15, 31, soft_cap};                      // front rows a geometry holds (16 / 4 / 1 scenarios per workgroup)
    for (int k = n - 1; k >= 0; --k) {
        if (!top[k] || S.top_task_of[k] >= 0) continue;
        Task t{};
        t.e = ssize(k);
        // a task takes at least top_mmin pivots (8; policy bits 54-59) where its front has room for them, and fills the front up to the soft cap
        // (more than 8 only while the front stays in class 3 -- 47 rows + the rhs column: one class-4 task makes its whole launch run the class-4 kernel, 1.0 us per
        // pivot step against 0.79)
        int cap = std::max(std::min(std::min(S.top_mmin, std::max(8, 47 - t.e)), TOP_FRONT_MAX - t.e), soft_cap - t.e), lowcap = 0;
        if (mid_mmin > 0) {
            const int need = std::max(1, std::min(tsub[k], mid_mmin));
            int geo = 2;
            for (int g = 0; g < 2 && geo == 2; ++g) if (gcap[g] - t.e >= need) geo = g;
            if (geo < 2) cap = gcap[geo] - t.e;
            lowcap = geo ? gcap[geo - 1] : 0;
        }
        const int id = (int)tasks.size();
        t.piv.push_back(k); S.top_task_of[k] = id;
        for (int q = k - 1; q > k - sub[k] && (int)t.piv.size() < cap; --q)
            if (top[q] && S.top_task_of[q] < 0 && S.top_task_of[parent(q)] == id && !(mid_strict && ssize(q) + 1 <= lowcap)) { t.piv.push_back(q); S.top_task_of[q] = id; }
        std::reverse(t.piv.begin(), t.piv.end());
        t.m = (int)t.piv.size(); t.parent = -1; t.level = 1; t.stack = -1;
        tasks.push_back(t);
    }
    const int nt = (int)tasks.size();
    // tasks were created from the root down: a child has a larger index than its parent
    for (int t = nt - 1; t >= 0; --t) {
        const int p = parent(tasks[t].root());
        if (p >= 0) { tasks[t].parent = S.top_task_of[p]; tasks[tasks[t].parent].kids.push_back(t); tasks[tasks[t].parent].level = std::max(tasks[tasks[t].parent].level, tasks[t].level + 1); }
    }
    long long stack = 0;
    for (Task& t : tasks) {
        const int fprime = t.m + t.e + 1;                        // front rows / columns + the rhs column
        t.cls = fprime <= 32 ? 2 : (fprime <= 48 ? 3 : 4);       // blocks per thread and dimension on the 16 x 16 thread grid
        t.lg = 0;
        if (mid_mmin > 0 && fprime <= 32) { t.lg = fprime <= 16 ? 4 : 2; t.cls = 4; }   // grouped: 16 / 4 scenarios per workgroup, 4 x 4 blocks per thread

        for (int k : t.piv) { const long long s = ssize(k); S.top_terms += s * (s + 1); }
    }
    // update blocks: e x (e + 1) blocks (update matrix | update vector) on the stack of the interleave class of the PARENT's workgroup
    // (blocks of different interleaves cannot share one address space: a class is a stack of its own)
    long long cstack[3] = {0, 0, 0};
    for (Task& t : tasks)
        if (t.e > 0) { long long& cs = cstack[t.parent >= 0 ? tasks[t.parent].lg >> 1 : 0]; t.stack = (int)cs; cs += (long long)t.e * (t.e + 1) * 4; }
    stack = cstack[0] + cstack[1] + cstack[2];
    if (stack >= (1LL << 31)) { S.top_task_of.assign(n, -1); return; }
    S.top_stack = stack;
    for (int c = 0; c < 3; ++c) S.top_stack_cls[c] = cstack[c];
    S.top_level = top_level;
    // launch order: level-major; ONE launch per level, compiled for the widest front of the level (a dependent launch costs more
    // than the registers a narrow front leaves unused in a wide kernel: the top levels hold a handful of tasks)
    std::vector<int> order(nt);
    std::iota(order.begin(), order.end(), 0);
    // (policy bit 3, large batches: a level's tasks of different classes go to launches of their own -- eight tasks x 512 scenarios are
    // 4 096 workgroups, and the six of them with small fronts then run the 71-register kernel, seven to a CU instead of four)
    const bool split = S.top_split != 0;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (tasks[a].level != tasks[b].level) return tasks[a].level < tasks[b].level;
        if ((tasks[a].lg > 0) != (tasks[b].lg > 0)) return tasks[a].lg > 0;             // the grouped tasks of a level: one launch, the longest first
        if (tasks[a].lg > 0 && tasks[a].m != tasks[b].m) return tasks[a].m > tasks[b].m;
        if (split && tasks[a].cls != tasks[b].cls) return tasks[a].cls < tasks[b].cls;
        return tasks[a].root() < tasks[b].root();
    });
    const bool sym = S.symmetric != 0;
    S.top_task.resize(nt);
    for (int oi = 0; oi < nt; ++oi) {
        const Task& t = tasks[order[oi]];
        const int m = t.m, e = t.e, f = m + e, fprime = f + 1;
        auto local = [&](const Task& tt, int pivot) {           // front index of a pivot in task tt, -1 if outside
            const auto it = std::lower_bound(tt.piv.begin(), tt.piv.end(), pivot);
            if (it != tt.piv.end() && *it == pivot) return (int)(it - tt.piv.begin());
            const int kl = tt.root();
            const int* b = S.u_col.data() + S.u_ptr[kl];
            const int* en = S.u_col.data() + S.u_ptr[kl + 1];
            const int* p = std::lower_bound(b, en, pivot);
            return (p != en && *p == pivot) ? tt.m + (int)(p - b) : -1;
        };
        Rec h{};
        const int base = (int)S.top_data.size();
        // entry map of the front, [f][f + 1]: what thread-slot (r, c) loads before and stores after the elimination
        //   -1 nothing (outside the pattern, or the update matrix / vector: starts from zero, leaves through the stack)
        //   -(2 + k) rhs row of pivot k (column f of the pivot's front row): loaded from and stored to the rhs storage
        //   entry | flags << 28: flag 1 starts from zero (fill-in that no bottom pivot touches), 2 read transposed (symmetric plans
        //   keep the upper triangle only: slot (r, c), r > c, reads entry (c, r)), 4 not stored by the slot's thread (the
        //   transposed copies; the diagonal blocks of the chain, which leave through the pivot wave in factorised form)
        auto bottom_terms = [&](int en) {                        // does the entry have a level item (build_tables)?
            for (int x = S.t_ptr[en]; x < S.t_ptr[en + 1]; ++x) if (S.top_task_of[S.e_col[S.t_a[x]]] < 0) return true;
            return false;
        };
        std::vector<int> emap((size_t)f * fprime, -1);
        auto put = [&](int en, int r, int c) {
            int flags = (!bottom_terms(en) && S.e_src[en] < 0) ? 1 : 0;
            if (r == c) flags |= 4;
            emap[(size_t)r * fprime + c] = en | flags << 28;
            if (sym && r != c) emap[(size_t)c * fprime + r] = en | (flags | 2 | 4) << 28;
        };
        std::vector<int> dent(m);
        for (int q = 0; q < m; ++q) {
            const int k = t.piv[q];
            dent[q] = S.diag[k];
            put(S.diag[k], q, q);
            emap[(size_t)q * fprime + f] = -(2 + k);
            for (int p = S.u_ptr[k]; p < S.u_ptr[k + 1]; ++p) {
                const int j = S.u_col[p], lj = local(t, j);
                put(S.u_ent[p], q, lj);
                if (!sym) put(find_in_row(S, j, k), lj, q);
            }
        }
        S.top_data.insert(S.top_data.end(), emap.begin(), emap.end());
        const int dent_off = (int)S.top_data.size() - base;
        S.top_data.insert(S.top_data.end(), dent.begin(), dent.end());
        // child records {stack offset, e_c, inv[f + 1]}: inv[x] = row / column of front index x in the child's update matrix,
        // -1 if the child does not reach it; inv[f] = e_c (the child's update vector is column e_c of its stack block)
        const int child_off = (int)S.top_data.size() - base;
        for (int c : t.kids) {
            const Task& ct = tasks[c];
            const int kl = ct.root();
            S.top_data.push_back(ct.stack); S.top_data.push_back(ct.e);
            std::vector<int> inv(fprime, -1);
            int idx = 0;
            for (int p = S.u_ptr[kl]; p < S.u_ptr[kl + 1]; ++p) inv[local(t, S.u_col[p])] = idx++;
            inv[f] = ct.e;
            S.top_data.insert(S.top_data.end(), inv.begin(), inv.end());
        }
        const int piv_off = (int)S.top_data.size() - base;
        S.top_data.insert(S.top_data.end(), t.piv.begin(), t.piv.end());
        h.w[0] = m; h.w[1] = e; h.w[2] = t.root(); h.w[3] = base; h.w[4] = t.stack; h.w[5] = (int)t.kids.size();
        h.w[6] = piv_off; h.w[7] = child_off; h.w[8] = dent_off; h.w[9] = t.cls; h.w[10] = t.level; h.w[11] = fprime;
        h.w[12] = t.parent >= 0 ? tasks[t.parent].lg : 0; h.w[13] = t.lg;
        h.w[14] = -1;
        if (want_jordan && e > 0) { h.w[14] = S.n_entries + S.n_jordan; S.n_jordan += m * e; }     // Jordan rows: [m][e] blocks behind the factor entries
        S.top_task[oi] = h;
        const int grouped = t.lg > 0 ? 1 : 0;
        if (S.top_launch.empty() || S.top_launch.back().level != t.level || S.top_launch.back().grouped != grouped ||
            (!grouped && split && S.top_launch.back().cls != t.cls))
            S.top_launch.push_back(TopLaunch{oi, 0, t.cls, t.level, grouped, (int)S.top_wgmap.size(), 0, 0});
        S.top_launch.back().ntasks++;
        S.top_launch.back().cls = std::max(S.top_launch.back().cls, t.cls);
        if (grouped) {
            const int blocks = 64 >> t.lg;                       // workgroups of this task per 64-scenario group
            for (int b = 0; b < blocks; ++b) S.top_wgmap.push_back(oi << 8 | b);
            S.top_launch.back().nwg += blocks;
        }
    }
    if (want_jordan) {
        // (symmetric plans too: a task mirrors its front from the upper entries, eliminates the full front and stores the upper part -- the
        // Jordan rows are upper blocks.  The in-task triangle then holds multipliers, not U: whoever reads U(k,i)' as Lh(i,k) of a TASK pivot
        // -- the forward elimination alone, the selected inverse -- factorises with Engine::jordan off.)
        bool plain = true;
        for (const Task& t : tasks) if (t.lg > 0) plain = false;
        if (plain) S.jordan = 1;
        else { S.n_jordan = 0; for (Rec& h : S.top_task) h.w[14] = -1; }
    }
    // top_task_of must name the header position
    std::vector<int> pos(nt);
    for (int oi = 0; oi < nt; ++oi) pos[order[oi]] = oi;
    for (int k = 0; k < n; ++k) if (S.top_task_of[k] >= 0) S.top_task_of[k] = pos[S.top_task_of[k]];
}

void build_tables(BlockSymbolic& S) {
    const int nE = S.n_entries, n = S.n;
    // factorisation + fused forward elimination
    std::vector<int> level(nE + n), work(nE + n);
    // symmetric matrices (policy bit 1): Lh(i,k) = U(k,i)' -- the lower entries are neither computed nor stored; a term
    // names U(k,i) with the TRANSPOSE bit (bit 30) where it would read Lh(i,k).  Half the update terms.
    const bool sym = S.symmetric != 0;
    auto lower_operand = [&](int e) { return sym ? (find_in_row(S, S.e_col[e], S.e_row[e]) | 1 << 30) : e; };
    // Top tasks (jg_symbolic.hpp): an item of a task-owned entry / rhs row keeps only the terms of BOTTOM pivots, stores the
    // partial sum raw, and is levelled on those terms alone; the terms of task pivots are executed inside the tasks.
    const bool has_top = !S.top_task.empty();
    auto in_top = [&](int k) { return has_top && S.top_task_of[k] >= 0; };
    auto owner = [&](int e) { return std::min(S.e_row[e], S.e_col[e]); };
    std::vector<int> ft_ptr(nE + n + 1, 0), ft_idx;              // scheduled term ids per item (entries: into t_*, rows: into l_*)
    ft_idx.reserve((size_t)S.n_terms + S.l_ptr[n]);
    for (int e = 0; e < nE; ++e) {
        const bool top = in_top(owner(e));
        int lev = top ? 0 : S.e_level[e];
        for (int t = S.t_ptr[e]; t < S.t_ptr[e + 1]; ++t) {
            if (top && in_top(S.e_col[S.t_a[t]])) continue;
            ft_idx.push_back(t);
            if (top) lev = std::max(lev, 1 + std::max(S.e_level[S.t_a[t]], std::max(S.e_level[S.t_d[t]], S.e_level[S.t_b[t]])));
        }
        ft_ptr[e + 1] = (int)ft_idx.size();
        work[e] = ft_ptr[e + 1] - ft_ptr[e];
        if (top && work[e] == 0 && !S.inplace && S.e_src[e] >= 0) lev = 1;           // copy of the caller's block
        level[e] = lev;
        if (S.inplace && work[e] == 0 && S.e_row[e] != S.e_col[e] && S.e_src[e] >= 0) level[e] = 0;   // already in place
        if (top && work[e] == 0 && (S.inplace || S.e_src[e] < 0)) level[e] = 0;      // in place already / starts from zero inside its task
        if (sym && S.e_row[e] > S.e_col[e]) level[e] = 0;
    }
    for (int r = 0; r < n; ++r) {
        const bool top = in_top(r);
        int lev = top ? 1 : S.y_level[r];
        for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) {
            const int c = S.l_col[p];
            if (top && in_top(c)) continue;
            ft_idx.push_back(p);
            if (top) lev = std::max(lev, 1 + std::max(S.y_level[c], std::max(S.e_level[S.l_ent[p]], S.e_level[S.diag[c]])));
        }
        ft_ptr[nE + r + 1] = (int)ft_idx.size();
        work[nE + r] = ft_ptr[nE + r + 1] - ft_ptr[nE + r];
        level[nE + r] = lev;
    }
    auto fill_fact = [&](int it, int sub, int wpi, int rpw, Rec* r) {
        int kind, id, src;
        if (it < nE) {
            kind = S.e_row[it] == S.e_col[it] ? 2 : (S.e_row[it] > S.e_col[it] ? 1 : 0);
            if (in_top(owner(it))) kind = 0;                     // partial sum of a task-owned entry: stored raw
            id = it; src = S.e_src[it] < 0 ? -1 : (S.inplace ? it : S.e_src[it]);
        } else {
            kind = 3; id = it - nE; src = S.perm[id];
        }
        for (int j = 0; j < rpw; ++j) { r[j].w[0] = kind; r[j].w[1] = id; r[j].w[2] = src; r[j].w[3] = 0; }
        int q = 0;
        for (int f = ft_ptr[it] + sub; f < ft_ptr[it + 1]; f += wpi, ++q) {
            const int t = ft_idx[f];
            Rec& x = r[q / FACT_T];
            const int s = 4 + 3 * (q % FACT_T);
            if (it < nE) { x.w[s] = lower_operand(S.t_a[t]); x.w[s + 1] = S.t_d[t]; x.w[s + 2] = S.t_b[t]; }
            else { x.w[s] = lower_operand(S.l_ent[t]); x.w[s + 1] = S.diag[S.l_col[t]]; x.w[s + 2] = S.l_col[t]; }
            x.w[3]++;
        }
    };
    // ---- the same items for a SINGLE instance (policy bit 60; jg_symbolic.hpp: SINGLE_FACT_LEVELS): a thread per item, two launches (built on a thread of its own, beside the other tables:
    // it only reads what the lists above hold)
    S.single_fact_ok = false; S.f_rec.clear(); S.f1_first.clear(); S.f1_wg.clear(); S.f2_first.clear(); S.n_f1_wg = 0;
    auto build_single_fact = [&] {
    if (S.want_single && has_top && !sym && S.inplace) {
        auto parent = [&](int k) { return S.u_ptr[k + 1] > S.u_ptr[k] ? S.u_col[S.u_ptr[k]] : -1; };
        auto item_pivot = [&](int it) { return it < nE ? owner(it) : it - nE; };
        // Records: the FIRST record of item j is record j (the items of a workgroup's level -- and the partial items behind them -- are read without an index in between);
        // the records an item of more than four terms continues in follow all first records, word 3 of the first = terms | first continuation record << 10.
        std::vector<int> order_items;                            // items in record order
        auto emit = [&](int it) { order_items.push_back(it); return (int)order_items.size() - 1; };
        bool ok = true;
        int maxlev = 0;
        std::vector<int> sub(n, 1);
        for (int k = 0; k < n; ++k) if (parent(k) >= 0) sub[parent(k)] += sub[k];
        // bottom subtrees in ascending order, their scheduled items by level
        std::vector<std::vector<int>> piv_items(n);
        for (int it = 0; it < nE + n; ++it) {
            if (level[it] <= 0) continue;
            const int p = item_pivot(it);
            if (in_top(p)) continue;
            piv_items[p].push_back(it);
            maxlev = std::max(maxlev, level[it]);
        }
        if (maxlev > SINGLE_FACT_LEVELS) ok = false;
        std::vector<std::vector<int>> wg_items;                  // per workgroup: items (any order; sorted by level below)
        int cur = SINGLE_FACT_ITEMS + 1;
        for (int k = 0; k < n && ok; ++k) {
            if (in_top(k)) continue;
            const int p = parent(k);
            if (p >= 0 && !in_top(p)) continue;                  // not the root of a bottom subtree
            int cnt = 0;
            for (int q = k - sub[k] + 1; q <= k; ++q) { if (in_top(q)) ok = false; cnt += (int)piv_items[q].size(); }
            if (cnt == 0) continue;
            if (cur + cnt > SINGLE_FACT_ITEMS) { wg_items.emplace_back(); cur = 0; }
            for (int q = k - sub[k] + 1; q <= k; ++q) for (int it : piv_items[q]) wg_items.back().push_back(it);
            cur += cnt;
        }
        if (ok) {
            for (std::vector<int>& items : wg_items) {
                std::stable_sort(items.begin(), items.end(), [&](int x, int y) { return level[x] < level[y]; });
                size_t x = 0;
                for (int l = 1; l <= SINGLE_FACT_LEVELS; ++l) {
                    S.f1_wg.push_back((int)S.f1_first.size());
                    while (x < items.size() && level[items[x]] == l) S.f1_first.push_back(emit(items[x++]));
                }
                S.f1_wg.push_back((int)S.f1_first.size());
            }
            S.n_f1_wg = (int)wg_items.size();
            for (int it = 0; it < nE + n; ++it) if (level[it] > 0 && in_top(item_pivot(it))) S.f2_first.push_back(emit(it));
            const int ni = (int)order_items.size();
            size_t total = (size_t)ni;
            for (int it : order_items) { const int nt = ft_ptr[it + 1] - ft_ptr[it]; if (nt > 1023) ok = false; total += (size_t)std::max(0, (nt + 3) / 4 - 1); }
            if (total >= (1u << 21)) ok = false;
            if (ok) {
                S.f_rec.assign(total, Rec{});
                size_t cont = (size_t)ni;
                std::vector<Rec> tmp;
                for (int j = 0; j < ni; ++j) {
                    const int it = order_items[j];
                    const int nt = ft_ptr[it + 1] - ft_ptr[it];
                    const int nrec = std::max(1, (nt + 3) / 4);
                    tmp.assign(nrec, Rec{});
                    fill_fact(it, 0, 1, nrec, tmp.data());       // (kind, id, src in every record; words 4 .. 15: up to four terms each)
                    S.f_rec[j] = tmp[0];
                    S.f_rec[j].w[3] = nt | (int)(nrec > 1 ? cont : 0) << 10;
                    for (int r = 1; r < nrec; ++r) S.f_rec[cont++] = tmp[r];
                }
            }
            S.single_fact_ok = ok;
        }
    }
    };
    S.n_sched_terms = S.top_terms;
    for (int it = 0; it < nE + n; ++it) if (level[it] > 0) S.n_sched_terms += work[it];
    // Item order inside a level: everything that becomes final with pivot p = min(row, col) together -- U(p, .) and y_p share
    // Lh(p, k) and D(k) of every term, Lh(., p) share U(k, p) and D(k) -- so the waves of a workgroup (and the workgroups that run
    // at the same moment on an XCD) ask for the same operand blocks while they are still in L1 / L2.  JG_ITEM_ORDER=0: heaviest first.
    std::vector<long long> loc(nE + n);
    for (int e = 0; e < nE; ++e) {
        const int r = S.e_row[e], c = S.e_col[e], p = std::min(r, c);
        loc[e] = ((long long)p << 33) | ((long long)(r > c ? 1 : 0) << 32) | (long long)(r > c ? r : c);
    }
    for (int r = 0; r < n; ++r) loc[nE + r] = ((long long)r << 33) | 0xffffffffLL;      // y_p right after U(p, .)
    const std::vector<long long>* locp = &loc;
    // the factorisation tables are two thirds of this function's time and independent of the others: they get a thread of their own
    // Plans with policy bit 50: the same items, levels and terms as TASKS (jg_symbolic.hpp) -- runs of items in the pivot order of their level
    // share a workgroup that stages their common operands, premultiplied by the pivot block, in LDS.
    auto build_fact_tasks = [&] {
        struct Term { int mem, slot, a, d, b; };                   // slot -1: no slot left, the term keeps the three-operand form
        struct TItem { int it, kind, side, wpi, rps; std::vector<Term> terms; };
        const int T = TASK_T, W = TASK_WAVES;
        const int rmax = std::max(1, S.task_rounds);
        int nlev = 0;
        for (int i = 0; i < nE + n; ++i) nlev = std::max(nlev, level[i]);
        std::vector<std::vector<int>> by(nlev + 1);
        for (int i = 0; i < nE + n; ++i) if (level[i] > 0) by[level[i]].push_back(i);
        S.fact_seg.clear(); S.fact_rec.clear(); S.n_fact_levels = 0; S.n_staged = 0; S.n_direct_terms = 0;
        // the levels are independent of each other: each is laid out into tables of its own (by several threads: the 512-scenario plan of a
        // 10 000-bus grid is 48 ms of this on one), stitched in level order afterwards
        struct LevelOut { std::vector<Segment> segs; std::vector<Rec> recs; long long staged = 0, direct = 0; };
        struct Scratch { std::vector<int> slot_stamp, slot_idx; int stamp = 0; };   // staged operand (an entry) -> slot of the task being filled
        std::vector<LevelOut> outs(nlev + 1);
        auto side_of = [&](int it) { return it < nE && S.e_row[it] > S.e_col[it] ? 1 : 0; };
        auto group_of = [&](int it) -> long long { return it < nE ? ((long long)std::min(S.e_row[it], S.e_col[it]) << 1 | side_of(it)) : ((long long)(it - nE) << 1); };
        constexpr int probe = 0;
        auto term_of = [&](int it, int f, int& a, int& d, int& b) {
            const int t = ft_idx[f];
            if (it < nE) { a = lower_operand(S.t_a[t]); d = S.t_d[t]; b = S.t_b[t]; if (probe == 2) { a = d; b = d; } }
            else { a = lower_operand(S.l_ent[t]); d = S.diag[S.l_col[t]]; b = S.l_col[t]; if (probe == 2) a = d; }
        };
        auto shares_of = [&](int it) { return std::min(W, pow2ceil(std::max(1, (work[it] + T - 1) / T))); };
        struct Task { std::vector<TItem> items; std::vector<int> skey, sd; int shares = 0; };
        auto do_level = [&](int l, Scratch& sc, LevelOut& out) {
            std::vector<int>& slot_stamp = sc.slot_stamp; std::vector<int>& slot_idx = sc.slot_idx; int& stamp = sc.stamp;
            std::vector<int>& its = by[l];
            if (its.empty()) return;
            std::stable_sort(its.begin(), its.end(), [&](int x, int y) { return loc[x] < loc[y]; });
            std::vector<Task> tasks;
            Task cur;
            ++stamp;
            auto close = [&] { if (!cur.items.empty()) { tasks.push_back(std::move(cur)); cur = Task(); ++stamp; } };
            auto new_keys = [&](size_t g0, size_t g1) {             // staged operands the items [g0, g1) would add to the current task
                int nk = 0;
                std::vector<int> seen;
                for (size_t x = g0; x < g1; ++x) {
                    const int it = its[x];
                    for (int f = ft_ptr[it]; f < (probe == 1 ? ft_ptr[it] : ft_ptr[it + 1]); ++f) {
                        int a, d, b; term_of(it, f, a, d, b);
                        const int key = (side_of(it) ? b : a) & 0x3fffffff;
                        if (slot_stamp[key] != stamp && slot_stamp[key] != -2 - stamp) { slot_stamp[key] = -2 - stamp; seen.push_back(key); ++nk; }
                    }
                }
                for (int k : seen) slot_stamp[k] = -1;
                return nk;
            };
            auto add_item = [&](int it) {
                TItem ti{};
                ti.it = it; ti.side = side_of(it);
                if (it < nE) { ti.kind = S.e_row[it] == S.e_col[it] ? 2 : 0; if (in_top(owner(it))) ti.kind = 0; } else ti.kind = 3;
                for (int f = ft_ptr[it]; f < (probe == 1 ? ft_ptr[it] : ft_ptr[it + 1]); ++f) {
                    Term tm{};
                    term_of(it, f, tm.a, tm.d, tm.b);
                    const int keyw = ti.side ? tm.b : tm.a;         // the shared operand (with its transpose bit)
                    const int key = keyw & 0x3fffffff;
                    tm.mem = ti.side ? tm.a : tm.b;
                    if (slot_stamp[key] == stamp) tm.slot = slot_idx[key];
                    else if ((int)cur.skey.size() < TASK_SLOTS) {
                        slot_stamp[key] = stamp; slot_idx[key] = (int)cur.skey.size(); tm.slot = slot_idx[key];
                        cur.skey.push_back(keyw | (ti.side ? 0 : 1 << 29)); cur.sd.push_back(tm.d);
                    } else tm.slot = -1;
                    ti.terms.push_back(tm);
                }
                ti.wpi = shares_of(it);
                cur.shares += ti.wpi;
                cur.items.push_back(std::move(ti));
            };
            for (size_t x = 0; x < its.size();) {
                size_t y = x;
                int sh = 0;
                while (y < its.size() && group_of(its[y]) == group_of(its[x])) sh += shares_of(its[y++]);
                const int nk = new_keys(x, y);
                if ((int)cur.skey.size() + nk <= TASK_SLOTS && cur.shares + sh <= W * rmax) { for (; x < y; ++x) add_item(its[x]); continue; }
                close();
                const int nk0 = new_keys(x, y);
                if (nk0 <= TASK_SLOTS && sh <= W * rmax) { for (; x < y; ++x) add_item(its[x]); continue; }
                for (; x < y; ++x) {                              // a row / column too large for one task: item by item
                    const int it = its[x];
                    if (!cur.items.empty() && ((int)cur.skey.size() + new_keys(x, x + 1) > TASK_SLOTS || cur.shares + shares_of(it) > W * rmax)) close();
                    add_item(it);
                }
            }
            close();
            // lay every task out: shares -> (wave, round); records
            struct Laid { int spw, rounds; std::vector<Rec> recs; };   // recs: [wave][rounds]
            std::vector<Laid> laid(tasks.size());
            for (size_t ti = 0; ti < tasks.size(); ++ti) {
                Task& tk = tasks[ti];
                for (TItem& it : tk.items) {                      // records a share needs: staged terms T per record, direct ones 4 per record
                    int rps = 1;
                    for (int sub = 0; sub < it.wpi; ++sub) {
                        int ns = 0, nd = 0;
                        for (size_t q = sub; q < it.terms.size(); q += it.wpi) (it.terms[q].slot >= 0 ? ns : nd)++;
                        rps = std::max(rps, (ns + T - 1) / T + (nd + TASK_DIRECT_T - 1) / TASK_DIRECT_T);
                    }
                    it.rps = rps;
                }
                std::vector<int> order(tk.items.size());
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                    if (tk.items[x].wpi != tk.items[y].wpi) return tk.items[x].wpi > tk.items[y].wpi;
                    return tk.items[x].rps > tk.items[y].rps;
                });
                int h[TASK_WAVES] = {0, 0, 0, 0, 0, 0, 0, 0};
                struct Place { int item, wave, round; };
                std::vector<Place> places;
                for (int oi : order) {
                    const TItem& it = tk.items[oi];
                    int best = 0, bh = 0x7fffffff;
                    for (int w0 = 0; w0 + it.wpi <= W; w0 += it.wpi) {
                        int mh = 0;
                        for (int w = w0; w < w0 + it.wpi; ++w) mh = std::max(mh, h[w]);
                        if (mh < bh) { bh = mh; best = w0; }
                    }
                    places.push_back(Place{oi, best, bh});
                    for (int w = best; w < best + it.wpi; ++w) h[w] = bh + it.rps;
                }
                int rounds = 1;
                for (int w = 0; w < W; ++w) rounds = std::max(rounds, h[w]);
                const int nslot = (int)tk.skey.size();
                const int spw = std::max(1, ((nslot + W - 1) / W + TASK_STAGE - 1) / TASK_STAGE);   // leading rounds whose records carry staging entries
                rounds = std::max(rounds, spw);
                std::vector<char> bar(rounds, 0);
                for (const Place& pl : places) if (tk.items[pl.item].wpi > 1) bar[pl.round + tk.items[pl.item].rps - 1] = 1;
                Laid& L = laid[ti];
                L.spw = spw; L.rounds = rounds;
                L.recs.assign((size_t)W * rounds, Rec{});
                for (int w = 0; w < W; ++w)
                    for (int r = 0; r < rounds; ++r) L.recs[(size_t)w * rounds + r].w[0] = 7 | (bar[r] ? TK_BAR : 0);
                for (int s = 0; s < nslot; ++s) {
                    const int w = s % W, q = s / W;
                    Rec& r = L.recs[(size_t)w * rounds + q / TASK_STAGE];
                    const int u = q % TASK_STAGE;
                    r.w[10 + 3 * u] = tk.skey[s]; r.w[11 + 3 * u] = tk.sd[s]; r.w[12 + 3 * u] = s;
                    r.w[3] = (u + 1) << 8;
                }
                out.staged += nslot;
                for (const Place& pl : places) {
                    const TItem& it = tk.items[pl.item];
                    const int id = it.it < nE ? it.it : it.it - nE;
                    const int src = it.it < nE ? (S.e_src[it.it] < 0 ? -1 : (S.inplace ? it.it : S.e_src[it.it])) : S.perm[id];
                    for (int sub = 0; sub < it.wpi; ++sub) {
                        std::vector<const Term*> st, dr;
                        for (size_t q = sub; q < it.terms.size(); q += it.wpi) (it.terms[q].slot >= 0 ? st : dr).push_back(&it.terms[q]);
                        size_t si = 0, di = 0;
                        for (int j = 0; j < it.rps; ++j) {
                            Rec& r = L.recs[(size_t)(pl.wave + sub) * rounds + pl.round + j];
                            int w0 = it.kind | (r.w[0] & TK_BAR) | (it.side ? TK_SIDE : 0) | sub << 8 | it.wpi << 12;
                            if (j == 0) w0 |= TK_FIRST;
                            if (j == it.rps - 1) w0 |= TK_LAST;
                            r.w[1] = id; r.w[2] = src;
                            int nt = 0;
                            if (si < st.size()) {
                                for (int q = 0; q < T && si < st.size(); ++q, ++si) { r.w[4 + q] = st[si]->mem | st[si]->slot << 24; ++nt; }
                            } else if (di < dr.size()) {
                                w0 |= TK_DIRECT;
                                for (int q = 0; q < TASK_DIRECT_T && di < dr.size(); ++q, ++di) { r.w[4 + 3 * q] = dr[di]->a; r.w[5 + 3 * q] = dr[di]->d; r.w[6 + 3 * q] = dr[di]->b; ++nt; out.direct++; }
                            }
                            r.w[3] |= nt;
                            r.w[0] = w0;
                        }
                    }
                }
            }
            // segments: the tasks of the level by shape, in task order (rec_base: relative to the level's records until the stitch)
            std::vector<char> done(tasks.size(), 0);
            for (size_t t0 = 0; t0 < tasks.size(); ++t0) {
                if (done[t0]) continue;
                Segment sg{};
                sg.rec_base = (int)out.recs.size(); sg.wpi = laid[t0].spw; sg.rpw = laid[t0].rounds; sg.level = l; sg.last = 0; sg.nchunks = 0; sg.items = 0;
                for (size_t t = t0; t < tasks.size(); ++t)
                    if (!done[t] && laid[t].spw == laid[t0].spw && laid[t].rounds == laid[t0].rounds) {
                        done[t] = 1;
                        out.recs.insert(out.recs.end(), laid[t].recs.begin(), laid[t].recs.end());
                        sg.nchunks++; sg.items += (int)tasks[t].items.size();
                    }
                out.segs.push_back(sg);
            }
            if (!out.segs.empty()) out.segs.back().last = 1;
        };
        // heaviest level first, levels handed out through a counter: the result does not depend on who laid out which level
        std::vector<int> lorder;
        for (int l = 1; l <= nlev; ++l) if (!by[l].empty()) lorder.push_back(l);
        std::stable_sort(lorder.begin(), lorder.end(), [&](int x, int y) { return by[x].size() > by[y].size(); });
        int nthr = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, lorder.size(), (size_t)std::max(1u, std::thread::hardware_concurrency() / 2), (size_t)(nE + n) / 8192 + 1}));
        if (knob_set("PLAN_THREADS")) nthr = std::max(1, std::min(16, knob("PLAN_THREADS", nthr)));     // tests: the tables must not depend on it
        std::atomic<size_t> next{0};
        auto worker = [&] {
            Scratch sc; sc.slot_stamp.assign(nE, -1); sc.slot_idx.assign(nE, 0);
            for (size_t q = next++; q < lorder.size(); q = next++) {
                const auto t0 = std::chrono::steady_clock::now();
                do_level(lorder[q], sc, outs[lorder[q]]);
                (void)t0;
            }
        };
        {
            std::vector<std::thread> pool;
            for (int t = 1; t < nthr; ++t) pool.push_back(spawn_or_run(worker));
            worker();
            for (std::thread& t : pool) if (t.joinable()) t.join();
        }
        for (int l = 1; l <= nlev; ++l) {
            LevelOut& o = outs[l];
            if (o.segs.empty()) continue;
            const int base = (int)S.fact_rec.size();
            for (Segment& sg : o.segs) { sg.rec_base += base; S.fact_seg.push_back(sg); }
            S.fact_rec.insert(S.fact_rec.end(), o.recs.begin(), o.recs.end());
            S.n_staged += o.staged; S.n_direct_terms += o.direct;
            ++S.n_fact_levels;
        }
    };
    const bool timing = knob_set("PLAN_TIMING");
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = tnow();
    std::thread fact_thread = spawn_or_run([&] {
        if (S.fact_tasks) build_fact_tasks();
        else build_replay(level, work, FACT_T, S.fact_seg, S.fact_rec, S.n_fact_levels, fill_fact, NoExtra(), 0, FACT_WAVES, locp);
        if (timing) fprintf(stderr, "[jg plan]   factorisation tables done at %6.1f ms\n", tnow() - tb0);
    });
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_fact{fact_thread};
    std::thread single_thread = spawn_or_run(build_single_fact);
    Join join_single{single_thread};
    // level 0 of a prefactor plan as tables of its own (for producers that deliver plain blocks): D(k) and y_k of the pivots
    // nobody updates -- and the forward-only tables: a thread of their own as well (round 4: the analysis is what a cold power flow waits for)
    S.pre_pivot.assign(n, 0); S.pre_seg.clear(); S.pre_rec.clear(); S.n_pre_levels = 0;
    std::thread side_thread = spawn_or_run([&] {
    if (S.prefactor) {
        std::vector<int> plevel(nE + n, 0), pwork(nE + n, 0);
        for (int k = 0; k < n; ++k)
            if (!in_top(k) && work[S.diag[k]] == 0 && S.e_level[S.diag[k]] == 0 && work[nE + k] == 0) {
                S.pre_pivot[k] = 1; plevel[S.diag[k]] = 1; plevel[nE + k] = 1;
            }
        build_replay(plevel, pwork, FACT_T, S.pre_seg, S.pre_rec, S.n_pre_levels, fill_fact, NoExtra(), 0, FACT_WAVES);
    }
    // forward elimination ALONE (factor once, solve many: fast decoupled power flow): the rhs rows only, levelled on
    // each other (every factor entry is final); all terms, no tasks
    {
        std::vector<int> flevel(nE + n, 0), fwork(nE + n, 0);
        for (int r = 0; r < n; ++r) {
            int l = 1;
            for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) l = std::max(l, flevel[nE + S.l_col[p]] + 1);
            flevel[nE + r] = l; fwork[nE + r] = S.l_ptr[r + 1] - S.l_ptr[r];
        }
        auto fill_fwd = [&](int it, int sub, int wpi, int rpw, Rec* r) {
            const int id = it - nE;
            for (int j = 0; j < rpw; ++j) { r[j].w[0] = 3; r[j].w[1] = id; r[j].w[2] = S.perm[id]; r[j].w[3] = 0; }
            int q = 0;
            for (int t = S.l_ptr[id] + sub; t < S.l_ptr[id + 1]; t += wpi, ++q) {
                Rec& x = r[q / FACT_T];
                const int s = 4 + 3 * (q % FACT_T);
                x.w[s] = lower_operand(S.l_ent[t]); x.w[s + 1] = S.diag[S.l_col[t]]; x.w[s + 2] = S.l_col[t];
                x.w[3]++;
            }
        };
        build_replay(flevel, fwork, FACT_T, S.fwd_seg, S.fwd_rec, S.n_fwd_levels, fill_fwd, NoExtra(), 0, FACT_WAVES);
    }
        if (timing) fprintf(stderr, "[jg plan]   pre + forward tables done at   %6.1f ms\n", tnow() - tb0);
    });
    Join join_side{side_thread};
    if (timing) fprintf(stderr, "[jg plan]   term lists of the items at      %6.1f ms\n", tnow() - tb0);
    // backward sweep: chains of a supernode go to ONE workgroup each (CHAIN_MAX_ROWS), the other rows stay wave records.
    // Built once for plain rows and, in Jordan plans, once more for Jordan rows (jg_symbolic.hpp): there a pivot of a top task is a plain
    // wave-record row over the EXTERNAL columns of its task (entries w14 + i e + c), never part of a chain.
    S.bwd_chain.clear();
    auto build_bwd = [&](bool jordan, std::vector<Segment>& out_seg, std::vector<Rec>& out_rec, int& out_levels, std::vector<int>* chain_level_out) {
        std::vector<int> jrow(n, -1), jroot(n, -1);              // Jordan: first block of the pivot's row, root of its task (whose U row names ext(task))
        if (jordan)
            for (const Rec& h : S.top_task) {
                const int* piv = S.top_data.data() + h.w[3] + h.w[6];
                for (int q = 0; q < h.w[0]; ++q) { jrow[piv[q]] = h.w[14] >= 0 ? h.w[14] + q * h.w[1] : 0; jroot[piv[q]] = h.w[2]; }
            }
        auto is_j = [&](int r) { return jroot[r] >= 0; };
        auto cols_of = [&](int r) { return is_j(r) ? jroot[r] : r; };    // the pivot whose U row lists the columns row r depends on
        std::vector<int> uw(n);
        for (int r = 0; r < n; ++r) { const int c = cols_of(r); uw[r] = S.u_ptr[c + 1] - S.u_ptr[c]; }
        std::vector<int> cstart, clen;                           // chains in ascending pivot order
        std::vector<int> chain_of(n);
        for (int k = 0; k < n;) {
            int e = k;
            if (!is_j(k))
                while (e + 1 < n && !is_j(e + 1) && e + 1 - k < CHAIN_MAX_ROWS && uw[e] >= 1 && S.u_col[S.u_ptr[e]] == e + 1 && uw[e] == uw[e + 1] + 1) ++e;
            if (uw[e] > CHAIN_MAX_EXT) e = k;                    // too many external columns for the LDS staging: plain rows
            for (int r = k; r <= e; ++r) chain_of[r] = (int)cstart.size();
            cstart.push_back(k); clen.push_back(e - k + 1);
            k = e + 1;
        }
        const int nc = (int)cstart.size();
        std::vector<int> clevel(nc, 1);
        for (int c = nc - 1; c >= 0; --c) {
            const int last = cols_of(cstart[c] + clen[c] - 1);
            for (int p = S.u_ptr[last]; p < S.u_ptr[last + 1]; ++p) clevel[c] = std::max(clevel[c], clevel[chain_of[S.u_col[p]]] + 1);
        }
        std::vector<int> row_level(n, 0);
        int max_level = 0;
        std::vector<std::vector<int>> chains_at;
        for (int c = 0; c < nc; ++c) {
            max_level = std::max(max_level, clevel[c]);
            if (chain_level_out) for (int r = cstart[c]; r < cstart[c] + clen[c]; ++r) (*chain_level_out)[r] = clevel[c];
            if (clen[c] == 1) row_level[cstart[c]] = clevel[c];
        }
        chains_at.assign(max_level + 1, {});
        for (int c = 0; c < nc; ++c) if (clen[c] > 1) chains_at[clevel[c]].push_back(c);
        build_replay(row_level, uw, BWD_T, out_seg, out_rec, out_levels, [&](int k, int sub, int wpi, int rpw, Rec* r) {
            for (int j = 0; j < rpw; ++j) { r[j].w[0] = k; r[j].w[1] = S.perm[k]; r[j].w[2] = S.diag[k]; r[j].w[3] = 0; }
            const int c = cols_of(k);
            int q = 0;
            for (int t = sub; t < uw[k]; t += wpi, ++q) {
                Rec& x = r[q / BWD_T];
                const int s = 4 + 2 * (q % BWD_T);
                x.w[s] = is_j(k) ? jrow[k] + t : S.u_ent[S.u_ptr[k] + t]; x.w[s + 1] = S.u_col[S.u_ptr[c] + t];
                x.w[3]++;
            }
        }, [&](int l, std::vector<Segment>& segs, std::vector<Rec>& recs) {
            static const bool no_small = knob("CHAIN_SMALL", 1) == 0;     // experiments: every chain a general task
            for (int small = 0; small < 2; ++small) {            // the general tasks (wpi 0), then the small ones (wpi -1, jg_symbolic.hpp)
            std::vector<int> cs;
            for (int c : chains_at[l]) if ((clen[c] <= CHAIN_SMALL_ROWS && !no_small) == (small != 0)) cs.push_back(c);
            if (cs.empty()) continue;
            Segment sg{};
            sg.rec_base = (int)recs.size(); sg.nchunks = (int)cs.size(); sg.wpi = small ? -1 : 0; sg.rpw = 1; sg.level = l; sg.last = 0; sg.items = 0;
            for (int c : cs) {
                const int b = clen[c], first = cstart[c], last = first + b - 1;
                const int nE = uw[last];
                Rec r{};
                r.w[0] = b; r.w[1] = nE; r.w[2] = (int)S.bwd_chain.size();
                int wpr = 1;
                while (wpr * 2 * b <= (small ? 8 : 16)) wpr *= 2;   // waves per row in the external phase
                r.w[3] = wpr;
                recs.push_back(r);
                sg.items += b;
                for (int p = 0; p < b; ++p) { S.bwd_chain.push_back(first + p); S.bwd_chain.push_back(S.perm[first + p]); S.bwd_chain.push_back(S.diag[first + p]); }
                for (int q = 0; q < nE; ++q) S.bwd_chain.push_back(S.u_col[S.u_ptr[last] + q]);
                for (int p = 0; p < b; ++p)
                    for (int q = 0; q < nE; ++q) S.bwd_chain.push_back(find_in_row(S, first + p, S.u_col[S.u_ptr[last] + q]));
                for (int p = 0; p < b; ++p)
                    for (int c2 = 0; c2 < b; ++c2) S.bwd_chain.push_back(c2 > p ? find_in_row(S, first + p, first + c2) : -1);
            }
            segs.push_back(sg);
            }
        }, max_level);
    };
    S.chain_level.assign(n, 0);
    build_bwd(false, S.bwd_seg, S.bwd_rec, S.n_bwd_levels, &S.chain_level);
    S.bwdj_seg.clear(); S.bwdj_rec.clear(); S.n_bwdj_levels = 0;
    if (timing) fprintf(stderr, "[jg plan]   backward tables done at        %6.1f ms\n", tnow() - tb0);
    if (S.jordan) build_bwd(true, S.bwdj_seg, S.bwdj_rec, S.n_bwdj_levels, nullptr);
    if (timing) fprintf(stderr, "[jg plan]   Jordan backward tables done at %6.1f ms\n", tnow() - tb0);
}

// Selected inverse Z = A^-1 on the pattern of the factor (Takahashi recursion, for a SYMMETRIC matrix: only the upper part
// and the diagonal are formed, Z(k,j) with k > j is read as Z(j,k)^T):
//   Z(i,j) = -D(i)^-1 * sum_{k in struct(i)} U(i,k) Z(k,j)            j in struct(i)           (level 2 l(i) - 1)
//   Z(i,i) =  D(i)^-1 * (I - sum_{k in struct(i)} U(i,k) Z(i,k)^T)                              (level 2 l(i))
// l(i) = backward-sweep level of pivot i (roots first).  Z lives in its own array with the factor's entry numbering.
// Record: w0 target entry, w1 D(i), w2 kind (0 off-diagonal, 1 diagonal), w3 terms, then (U entry, Z entry | transpose << 30) x 6.
int entry_of(const BlockSymbolic& S, int r, int c) { return find_in_row(S, r, c); }

void build_selected_inverse(BlockSymbolic& S) {
    if (!S.sel_seg.empty()) return;
    const int n = S.n;
    struct Item { int pivot, q; };                      // q = position in the U row of the pivot; -1 = the diagonal
    std::vector<Item> items;
    std::vector<int> level, work;
    for (int i = 0; i < n; ++i) {
        const int s = S.u_ptr[i + 1] - S.u_ptr[i];
        for (int q = 0; q < s; ++q) { items.push_back(Item{i, q}); level.push_back(2 * S.bwd_level[i] - 1); work.push_back(s); }
        items.push_back(Item{i, -1}); level.push_back(2 * S.bwd_level[i]); work.push_back(s);
    }
    build_replay(level, work, BWD_T, S.sel_seg, S.sel_rec, S.n_sel_levels, [&](int x, int sub, int wpi, int rpw, Rec* r) {
        const int i = items[x].pivot, q = items[x].q;
        const int target = q < 0 ? S.diag[i] : S.u_ent[S.u_ptr[i] + q];
        const int j = q < 0 ? i : S.u_col[S.u_ptr[i] + q];
        for (int t = 0; t < rpw; ++t) { r[t].w[0] = target; r[t].w[1] = S.diag[i]; r[t].w[2] = q < 0 ? 1 : 0; r[t].w[3] = 0; }
        int c = 0;
        for (int p = S.u_ptr[i] + sub; p < S.u_ptr[i + 1]; p += wpi, ++c) {
            const int k = S.u_col[p];
            int z;
            if (q < 0) z = S.u_ent[p] | 1 << 30;                                        // Z(k,i) = Z(i,k)^T
            else if (k == j) z = S.diag[j];
            else if (k < j) z = find_in_row(S, k, j);
            else z = find_in_row(S, j, k) | 1 << 30;
            Rec& rec = r[c / BWD_T];
            rec.w[4 + 2 * (c % BWD_T)] = S.u_ent[p];
            rec.w[5 + 2 * (c % BWD_T)] = z;
            rec.w[3]++;
        }
    }, NoExtra(), 0, FACT_WAVES);
}

// Tables of the shared-factor solve (jg_symbolic.hpp: CompTables).  Forward levels follow the elimination tree (a row's lower entries are
// its descendants), so "forward level > split" is closed under taking ancestors: the top is a union of root paths and no bottom row
// depends on it; a top row's partial sum is levelled behind the bottom rows it reads.
void build_comp_tables(const BlockSymbolic& S, int top_cap, CompTables& out) {
    const int n = S.n;
    out = CompTables();
    std::vector<int> flev(n, 1);
    int maxlev = 1;
    for (int r = 0; r < n; ++r) {
        for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) flev[r] = std::max(flev[r], flev[S.l_col[p]] + 1);
        maxlev = std::max(maxlev, flev[r]);
    }
    std::vector<int> count(maxlev + 2, 0);
    for (int r = 0; r < n; ++r) count[flev[r]]++;
    int split = maxlev;                                           // pivots with flev > split form the top
    if (top_cap >= 0) {
        int above = 0;
        while (split > 1 && above + count[split] <= top_cap) { above += count[split]; --split; }
    }
    out.split = split;
    out.tpos.assign(n, -1);
    for (int r = 0; r < n; ++r) if (flev[r] > split) { out.tpos[r] = (int)out.top.size(); out.top.push_back(r); }
    out.n_top = (int)out.top.size();
    const std::vector<int>& tpos = out.tpos;
    // forward: bottom rows with all their terms; top rows with their bottom terms, one level behind the last bottom row they read
    {
        std::vector<int> level(n, 0), work(n, 0);
        for (int r = 0; r < n; ++r) {
            if (tpos[r] < 0) { level[r] = flev[r]; work[r] = S.l_ptr[r + 1] - S.l_ptr[r]; continue; }
            int l = 1, w = 0;
            for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) if (tpos[S.l_col[p]] < 0) { l = std::max(l, flev[S.l_col[p]] + 1); ++w; }
            level[r] = l; work[r] = w;
        }
        build_replay(level, work, COMP_T, out.fwd_seg, out.fwd_rec, out.n_fwd_levels, [&](int k, int sub, int wpi, int rpw, Rec* r) {
            for (int j = 0; j < rpw; ++j) { r[j].w[0] = tpos[k] < 0 ? k : n + tpos[k]; r[j].w[1] = S.perm[k]; r[j].w[2] = -1; r[j].w[3] = 0; }
            int q = 0, seen = 0;
            for (int p = S.l_ptr[k]; p < S.l_ptr[k + 1]; ++p) {
                if (tpos[k] >= 0 && tpos[S.l_col[p]] >= 0) continue;                 // a top row leaves its top terms to the dense inverse
                if (seen++ % wpi != sub) continue;
                Rec& x = r[q / COMP_T];
                const int s = 4 + 2 * (q % COMP_T);
                x.w[s] = S.l_ent[p]; x.w[s + 1] = S.l_col[p];
                x.w[3]++; ++q;
            }
        }, NoExtra(), 0, 8);
    }
    // backward: bottom rows only (the top writes x_T itself), levelled on the bottom columns they read
    {
        std::vector<int> level(n, 0), work(n, 0);
        for (int r = n - 1; r >= 0; --r) {
            if (tpos[r] >= 0) continue;
            int l = 1;
            for (int p = S.u_ptr[r]; p < S.u_ptr[r + 1]; ++p) if (tpos[S.u_col[p]] < 0) l = std::max(l, level[S.u_col[p]] + 1);
            level[r] = l; work[r] = S.u_ptr[r + 1] - S.u_ptr[r];
        }
        build_replay(level, work, COMP_T, out.bwd_seg, out.bwd_rec, out.n_bwd_levels, [&](int k, int sub, int wpi, int rpw, Rec* r) {
            for (int j = 0; j < rpw; ++j) { r[j].w[0] = k; r[j].w[1] = S.perm[k]; r[j].w[2] = S.diag[k]; r[j].w[3] = 0; }
            int q = 0;
            for (int p = S.u_ptr[k] + sub; p < S.u_ptr[k + 1]; p += wpi, ++q) {
                Rec& x = r[q / COMP_T];
                const int s = 4 + 2 * (q % COMP_T);
                x.w[s] = S.u_ent[p]; x.w[s + 1] = S.u_col[p];
                x.w[3]++;
            }
        }, NoExtra(), 0, 8);
    }
}

// Tables of the single-instance backward sweep (jg_symbolic.hpp: SingleTables).
void build_single_tables(const BlockSymbolic& S, SingleTables& T) {
    T = SingleTables();
    const int n = S.n;
    if (!S.jordan || S.top_task.empty() || S.symmetric) return;
    std::vector<int> jrow(n, -1), jroot(n, -1), tlev(n, 0);
    int max_tl = 0;
    for (const Rec& h : S.top_task) max_tl = std::max(max_tl, h.w[10]);
    for (const Rec& h : S.top_task) {
        const int* piv = S.top_data.data() + h.w[3] + h.w[6];
        for (int q = 0; q < h.w[0]; ++q) { jrow[piv[q]] = h.w[14] >= 0 ? h.w[14] + q * h.w[1] : 0; jroot[piv[q]] = h.w[2]; tlev[piv[q]] = max_tl - h.w[10]; }
    }
    // ---- top rows by level (root task = level 0); a row's columns are the pivots of ancestor tasks, i.e. of earlier levels
    std::vector<int> rows;
    for (int k = 0; k < n; ++k) if (jroot[k] >= 0) rows.push_back(k);
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return tlev[a] < tlev[b]; });
    T.n_top = (int)rows.size();
    T.n_top_levels = max_tl;
    std::vector<int> slot(n, -1);
    for (int i = 0; i < T.n_top; ++i) slot[rows[i]] = i;
    T.t_level.assign(max_tl + 1, 0);
    for (int k : rows) T.t_level[tlev[k] + 1]++;
    for (int l = 0; l < max_tl; ++l) T.t_level[l + 1] += T.t_level[l];
    std::vector<int> slist(n, -1);                               // root pivot of a task -> first slot of the task's column list (one list per task: its rows share ext(task))
    for (int k : rows) {
        const int root = jroot[k], e = S.u_ptr[root + 1] - S.u_ptr[root];
        if (slist[root] < 0) {
            slist[root] = (int)T.t_term.size();
            for (int t = 0; t < e; ++t) {
                const int c = S.u_col[S.u_ptr[root] + t];
                if (slot[c] < 0 || tlev[c] >= tlev[k]) return;  // (cannot happen: ext(task) lies in ancestor tasks)
                T.t_term.push_back(slot[c]);
            }
        }
        T.t_row.push_back(k); T.t_row.push_back(S.perm[k]); T.t_row.push_back(S.diag[k]); T.t_row.push_back(e);
        T.t_ptr.push_back(e > 0 ? jrow[k] - S.n_entries : 0); T.t_ptr.push_back(slist[root]);
    }
    if (T.t_term.empty()) T.t_term.push_back(0);
    // terms as lanes: the compact blocks of a level must follow each other in (row, column) order (levels themselves sit in reverse order: the tasks were numbered leaves first)
    {
        T.flat_ok = true;
        T.t_jb.assign(2 * (size_t)max_tl, 0);                   // per level: first compact block, terms
        T.t_cslot.assign((size_t)std::max(S.n_jordan, 1), 0);
        long long covered = 0;
        for (int l = 0; l < max_tl; ++l) {
            T.max_level_rows = std::max(T.max_level_rows, T.t_level[l + 1] - T.t_level[l]);
            int first = -1, terms = 0;
            for (int i = T.t_level[l]; i < T.t_level[l + 1]; ++i) {
                const int e = T.t_row[4 * i + 3], jb = T.t_ptr[2 * i], sl = T.t_ptr[2 * i + 1];
                T.t_toff.push_back(terms);
                if (e == 0) continue;
                if (first < 0) first = jb;
                if (jb != first + terms || jb + e > S.n_jordan) { T.flat_ok = false; break; }
                for (int t = 0; t < e; ++t) T.t_cslot[jb + t] = T.t_term[sl + t];
                terms += e;
            }
            if (!T.flat_ok) break;
            T.t_jb[2 * l] = std::max(first, 0); T.t_jb[2 * l + 1] = terms;
            T.max_level_terms = std::max(T.max_level_terms, terms);
            covered += terms;
        }
        if (covered != S.n_jordan) T.flat_ok = false;            // (every block of the Jordan rows is some row's term)
        if ((int)T.t_toff.size() != T.n_top) T.flat_ok = false;
    }
    // ---- bottom rows: whole subtrees per workgroup
    auto is_top = [&](int k) { return jroot[k] >= 0; };
    auto parent = [&](int k) { return S.u_ptr[k + 1] > S.u_ptr[k] ? S.u_col[S.u_ptr[k]] : -1; };
    std::vector<int> sub(n, 1);
    for (int k = 0; k < n; ++k) if (parent(k) >= 0) sub[parent(k)] += sub[k];
    std::vector<int> blev(n, 0);
    for (int k = n - 1; k >= 0; --k) {
        if (is_top(k)) continue;
        for (int p = S.u_ptr[k]; p < S.u_ptr[k + 1]; ++p) { const int c = S.u_col[p]; if (!is_top(c)) blev[k] = std::max(blev[k], blev[c] + 1); }
    }
    // subtree roots in ascending order; a subtree = pivots root - sub[root] + 1 .. root, all below the top
    std::vector<std::pair<int, int>> trees;                      // (first pivot, rows)
    for (int k = 0; k < n; ++k) {
        if (is_top(k)) continue;
        const int p = parent(k);
        if (p >= 0 && !is_top(p)) continue;
        for (int q = k - sub[k] + 1; q <= k; ++q) if (is_top(q)) return;   // (a top pivot below a bottom one: the level rule forbids it)
        if (sub[k] > SINGLE_BOTTOM_ROWS) return;                  // a subtree one workgroup cannot hold: the level launches stay
        trees.push_back({k - sub[k] + 1, sub[k]});
    }
    std::vector<int> wg_of(n, -1), thr_of(n, -1);
    T.b_wg.clear();
    int cur_rows = SINGLE_BOTTOM_ROWS + 1;
    std::vector<std::vector<int>> wg_rows;
    for (const auto& tr : trees) {
        if (cur_rows + tr.second > SINGLE_BOTTOM_FILL) { wg_rows.emplace_back(); cur_rows = 0; }
        for (int q = tr.first; q < tr.first + tr.second; ++q) { wg_of[q] = (int)wg_rows.size() - 1; thr_of[q] = cur_rows++; wg_rows.back().push_back(q); }
    }
    T.n_wg = (int)wg_rows.size();
    for (int w = 0; w < T.n_wg; ++w) {
        int levels = 0;
        T.b_wg.push_back((int)T.b_row.size() / 6);
        for (int k : wg_rows[w]) {
            levels = std::max(levels, blev[k] + 1);
            const int nt = S.u_ptr[k + 1] - S.u_ptr[k];
            T.b_row.push_back(k); T.b_row.push_back(S.perm[k]); T.b_row.push_back(S.diag[k]); T.b_row.push_back(nt);
            T.b_row.push_back((int)T.b_term.size() / 2); T.b_row.push_back(blev[k]);
            for (int p = S.u_ptr[k]; p < S.u_ptr[k + 1]; ++p) {
                const int c = S.u_col[p];
                if (!is_top(c) && wg_of[c] != w) return;           // (cannot happen: the columns of a row are its ancestors)
                T.b_term.push_back(S.u_ent[p]); T.b_term.push_back(is_top(c) ? c : -(1 + thr_of[c]));
            }
        }
        T.b_wg.push_back(levels);
        T.b_levels = std::max(T.b_levels, levels);
    }
    T.b_wg.push_back((int)T.b_row.size() / 6); T.b_wg.push_back(0);
    T.n_bottom = (int)T.b_row.size() / 6;
    if (T.n_top + T.n_bottom != n) return;
    T.ok = true;
}

int analyze(int n, const int* rowptr, const int* col, long long policy64, BlockSymbolic& S) {
    S = BlockSymbolic();
    const int policy = (int)(policy64 & 0x7fffffff);
    S.inplace = policy & 1;
    constexpr int TOP_LEVEL_MIN = 6, TOP_NARROW = 384, TOP_FRONT_SOFT = 24;
    S.symmetric = (policy >> 1) & 1;
    S.prefactor = ((policy >> 2) & 1) && S.inplace;
    S.top_split = (policy >> 3) & 1;
    S.jordan = (int)((policy64 >> 49) & 1);                       // a request here; build_top grants it
    S.fact_tasks = (int)((policy64 >> 50) & 1);
    S.task_rounds = (int)((policy64 >> 51) & 7);
    S.want_single = (int)((policy64 >> 60) & 1);
    S.top_mmin = (int)((policy64 >> 54) & 0x3f);
    if (S.top_mmin <= 0) S.top_mmin = 8;
    if (S.task_rounds <= 0) S.task_rounds = 3;
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
    const bool timing = knob_set("PLAN_TIMING");
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = tnow();
    auto lap = [&](const char* what) { if (timing) { const double t = tnow(); fprintf(stderr, "[jg plan] %-28s %8.1f ms\n", what, t - t0); t0 = t; } };
    if (!elimination_order(n, std::move(adj), S.perm, strct)) return 3;     // JG_ORDER_CHECK only: an incremental fill disagreed with its recount
    lap("elimination order");
    postorder(S.perm, strct);
    lap("postorder");
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

    lap("  row patterns");
    // source positions in the caller's block CSR
    S.e_src.assign(S.n_entries, -1);
    S.src_entry.assign(rowptr[n], -1);
    for (int i = 0; i < n; ++i)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            int e = find_in_row(S, S.iperm[i], S.iperm[col[p]]);
            if (e < 0) return 1;
            if (S.e_src[e] >= 0) return 1;                    // duplicate block in the input pattern
            S.e_src[e] = p;
            S.src_entry[p] = e;
        }

    lap("  source map");
    // update terms: pivot k contributes -L(i,k) U(k,j) to every (i,j) in struct(k)^2.  The entry of (i, j) is looked up ONCE per term
    // (pivot-major list tgt: the count, the term lists and the dependency levels below all walk it) and by a merge of the sorted
    // row i with the sorted struct(k), not by a binary search per pair; lrow / urow: the entries L(s_a, k), U(k, s_a) of pivot k.
    std::vector<int> tgt, lrow, urow, sptr(n + 1, 0);
    for (int k = 0; k < n; ++k) sptr[k + 1] = sptr[k] + (int)strct[k].size();
    lrow.resize(sptr[n]); urow.resize(sptr[n]);
    {
        size_t nt = 0;
        for (int k = 0; k < n; ++k) nt += strct[k].size() * strct[k].size();
        tgt.resize(nt);
        size_t q = 0;
        for (int k = 0; k < n; ++k) {
            const std::vector<int>& s = strct[k];
            for (size_t a = 0; a < s.size(); ++a) {
                const int i = s[a];
                int e = S.row_ptr[i];
                const int e1 = S.row_ptr[i + 1];
                while (S.e_col[e] != k) ++e;                   // L(i, k): k precedes every member of struct(k) in row i
                lrow[sptr[k] + a] = e;
                for (size_t b = 0; b < s.size(); ++b) {        // struct(k) is inside row i (a clique): one forward scan
                    while (e < e1 && S.e_col[e] != s[b]) ++e;
                    tgt[q++] = e;
                }
            }
            int e = S.diag[k];
            for (size_t a = 0; a < s.size(); ++a) { while (S.e_col[e] != s[a]) ++e; urow[sptr[k] + a] = e; }
        }
    }
    lap("  term targets");
    S.t_ptr.assign(S.n_entries + 1, 0);
    for (int t : tgt) S.t_ptr[t + 1]++;
    for (int e = 0; e < S.n_entries; ++e) S.t_ptr[e + 1] += S.t_ptr[e];
    S.n_terms = S.t_ptr[S.n_entries];
    S.t_a.assign((size_t)S.n_terms, 0);
    S.t_d.assign((size_t)S.n_terms, 0);
    S.t_b.assign((size_t)S.n_terms, 0);
    {
        std::vector<int> fill(S.t_ptr.begin(), S.t_ptr.end() - 1);
        size_t q = 0;
        for (int k = 0; k < n; ++k) {
            const int ns = sptr[k + 1] - sptr[k], dk = S.diag[k];
            const int* lid = lrow.data() + sptr[k];
            const int* uid = urow.data() + sptr[k];
            for (int a = 0; a < ns; ++a)
                for (int b = 0; b < ns; ++b) {
                    const int f = fill[tgt[q++]]++;
                    S.t_a[f] = lid[a]; S.t_d[f] = dk; S.t_b[f] = uid[b];
                }
        }
    }

    lap("  term lists");
    // entry dependency levels (pivots ascending: all sources of pivot-k entries have smaller pivots);
    // D(k), U(k,.) and Lh(.,k) have no mutual dependency, so one level per pivot "generation"
    S.e_level.assign(S.n_entries, 0);
    {
        std::vector<int> acc(S.n_entries, 0);
        size_t q = 0;
        for (int k = 0; k < n; ++k) {
            const int ns = sptr[k + 1] - sptr[k], d = S.diag[k];
            const int* lid = lrow.data() + sptr[k];
            const int* uid = urow.data() + sptr[k];
            // level 0 (policy bit 2): entries the producer leaves final -- no update terms and present in its pattern
            auto lvl = [&](int e) { return (S.prefactor && S.t_ptr[e + 1] == S.t_ptr[e] && S.e_src[e] >= 0) ? 0 : acc[e] + 1; };
            S.e_level[d] = lvl(d);
            for (int a = 0; a < ns; ++a) { S.e_level[uid[a]] = lvl(uid[a]); S.e_level[lid[a]] = lvl(lid[a]); }
            for (int a = 0; a < ns; ++a) {
                const int li = std::max(S.e_level[lid[a]], S.e_level[d]);
                for (int b = 0; b < ns; ++b) {
                    const int t = tgt[q++];
                    acc[t] = std::max(acc[t], std::max(li, S.e_level[uid[b]]));
                }
            }
        }
    }

    lap("  entry levels");
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
    S.y_level.assign(n, S.prefactor ? 0 : 1);                 // a row without lower entries: y_k = rhs_k, written by the producer
    S.bwd_level.assign(n, 1);
    for (int r = 0; r < n; ++r)
        for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) {
            const int c = S.l_col[p];
            S.y_level[r] = std::max(S.y_level[r], 1 + std::max(S.y_level[c], std::max(S.e_level[S.l_ent[p]], S.e_level[S.diag[c]])));
        }
    for (int r = n - 1; r >= 0; --r)
        for (int p = S.u_ptr[r]; p < S.u_ptr[r + 1]; ++p) S.bwd_level[r] = std::max(S.bwd_level[r], S.bwd_level[S.u_col[p]] + 1);

    {
        int top_level = (policy >> 8) & 0xff, soft = (policy >> 16) & 0xff;
        if (knob_set("TOP_LEVEL")) top_level = knob("TOP_LEVEL", top_level);
        if (top_level == 0) {
            // default: the multifrontal top starts where the level schedule gets narrow -- the first dependency level from which
            // no level holds more than TOP_NARROW items (entries + rhs rows), but not below level TOP_LEVEL_MIN.  Measured on
            // ACTIVSg10k (84 levels): top from level 27-30 is the optimum at 64 AND at 512 scenarios (from level 12: +5 % at 64,
            // +35 % at 512 -- a task needs a workgroup per scenario, the level kernel only a wave per 64).
            int narrow = ((policy >> 24) & 0x7f) * 8;             // the owner knows its batch: a task costs a workgroup per scenario
            if (narrow == 0) narrow = TOP_NARROW;
            if (narrow == 127 * 8) narrow = 0x7fffffff;           // 127: no item limit (the pivots-per-level limit decides)
            int nlev = 0;
            for (int e = 0; e < S.n_entries; ++e) nlev = std::max(nlev, S.e_level[e]);
            for (int r = 0; r < n; ++r) nlev = std::max(nlev, S.y_level[r]);
            int chains = (policy >> 4) & 0xf;                     // ... and at most this many pivots (parallel chains) per level (0: any;
            if (chains >= 13) chains = chains == 13 ? 18 : (chains == 14 ? 24 : 36);   // 13 / 14 / 15 stand for 18 / 24 / 36)
            // ONE scenario (policy bit 60): a level below the top costs its share of k_fact1_bottom / k_bwd1_bottom (~4 + 2 us), a task level ~20 us whatever it holds, so the
            // wide levels stay below the top -- it starts where no level holds more than 150 pivots (ACTIVSg10k: level 6 instead of 5, 0.950 -> 0.863 ms per solve; 9241-bus
            // grid: level 8, 0.972 -> 0.923; profiles/r06_top_level_sweep.txt)
            if (S.want_single && chains == 0) chains = 150;
            std::vector<int> cnt(nlev + 2, 0), piv(nlev + 2, 0);
            for (int e = 0; e < S.n_entries; ++e) if (!(S.symmetric && S.e_row[e] > S.e_col[e])) cnt[S.e_level[e]]++;
            for (int r = 0; r < n; ++r) { cnt[S.y_level[r]]++; piv[S.e_level[S.diag[r]]]++; }
            top_level = nlev + 1;
            const int level_min = TOP_LEVEL_MIN - (S.prefactor ? 1 : 0);    // a prefactor plan numbers the same levels one lower: same pivots in the top
            while (top_level > level_min && cnt[top_level - 1] <= narrow && (chains == 0 || piv[top_level - 1] <= chains)) --top_level;
            if (top_level > nlev - 2) top_level = NO_TOP_LEVEL;  // nothing worth a task
        } else if (top_level >= 255) top_level = NO_TOP_LEVEL;   // policy / environment: no top by level
        if (soft <= 0) soft = TOP_FRONT_SOFT;
        int struct_min = 0, mid_mmin = 0, mid_strict = (int)((policy64 >> 48) & 1);
        if (const int mid = (int)((policy64 >> 32) & 0xff)) { struct_min = mid; mid_mmin = (int)((policy64 >> 40) & 0xff); if (!mid_mmin) mid_mmin = 6; }
        lap("fill pattern, terms, levels");
        build_top(S, top_level, std::min(soft, TOP_FRONT_MAX), struct_min, mid_mmin, mid_strict);
        lap("top tasks");
    }
    if (S.n_entries + S.n_jordan >= (1 << 24)) S.fact_tasks = 0;   // a task term packs (entry | slot << 24)
    build_tables(S);
    lap("replay tables");
    return 0;
}

}  // namespace jg

// own_bvh.h -- a binary BVH of the backend's OWN over the reference's LEAVES (host code; rt_scene_upload, round 4).
//
// Why a second tree is legal for shadow rays (DESIGN.md "a tree of its own"): the reference's any-hit query
// (trace_bvh.cl -DSHADOW_RAYS, :107-109,164-167) never changes t_max, so its verdict is
//     OR over the leaves L whose RayBounds passes (then every ancestor's passes: bounds are exact unions, bvh.hpp:73)
//        OR over the triangles T of L of RayTriangle(T)
// -- a boolean that does not depend on the order in which leaves are reached nor on what sits ABOVE the leaves.  Any tree
// whose leaves are exactly the reference's leaves (same exact box, same triangles in the same array) and whose interior
// boxes contain the boxes below them gives the same boolean when interior culling is conservative and every leaf is
// decided by the reference's own expression on its exact box -- which is what k_trace_w4 already does.  So this file only
// has to produce a BETTER binary tree over those leaves; build_wide_bvh then folds it into 4-wide records exactly as it
// folds the reference's tree, and the kernel does not change at all.
//
// What "better" means here, and what src/bvh.cpp:67-221 (the builder this one is not bound to) leaves on the table:
//  * the reference splits along ONE axis (the longest of the centroid bounds) at one of 11 bucket borders; this builder
//    sweeps EVERY border between two sorted centroids on all three axes (binned with 256 / 64 bins above 768 leaves);
//  * the cost of a box is the probability that a ray of the population the tree serves crosses it.  For rays of every
//    direction that is the surface area; shadow rays towards a directional light all share ONE direction d, and a line
//    of direction d crosses a box with probability proportional to its PROJECTED area |d.x| dy dz + |d.y| dz dx +
//    |d.z| dx dy.  The metric is a mixture: one projected-area term per directional light, one isotropic term (half the
//    average of the projected area over all directions = (dx dy + dy dz + dz dx) / 2) per point light or per unit of
//    `iso_weight` (Metric below) -- with no directional term it is the ordinary surface-area heuristic.
// The same builder with the isotropic metric gives the closest-hit tree of the opt-in tolerance mode
// (RT_CTX_OPT_CLOSEST_TREE): near child first by the sign of the ray along each node's split axis, like the reference, but
// on this topology -- identical results except where two candidate hits tie to within the rounding of the ray-triangle
// test (DESIGN.md has the argument and the measured differing-pixel counts).
//
// Output: rt_bvh_node[] in the reference's own linear layout (bvh.cpp:223-245: first child at i + 1, second child at
// `offset`, split axis in the low half of num_primitives_axis), leaves copied from the reference's array.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "rt_types.h"

namespace ownbvh
{
struct Metric
{
    double iso = 1.0;                       // weight of the isotropic term
    std::vector<std::array<double, 3>> dirs; // |d| components of unit directions, one projected-area term each
    double of(const float mn[3], const float mx[3]) const
    {
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        double m = iso * 0.5 * (dx * dy + dy * dz + dz * dx);
        for (const auto& d : dirs) m += d[0] * dy * dz + d[1] * dz * dx + d[2] * dx * dy;
        return m;
    }
};

struct Prim { float mn[3], mx[3], c[3]; uint32_t leaf; };

struct Builder
{
    const rt_bvh_node* ref;
    Metric metric;
    std::vector<Prim> prims;
    std::vector<rt_bvh_node> out;
    struct Task { uint32_t b, e, pos; };
    uint32_t grain = 16384;                 // a range of at most this many leaves is one task: its whole subtree, built by one thread
    unsigned n_threads = 1;
    const std::atomic<bool>* cancel = nullptr;
    bool cancelled() const { return cancel && cancel->load(std::memory_order_relaxed); }

    static void grow(float mn[3], float mx[3], const Prim& p)
    {
        for (int a = 0; a < 3; ++a) { mn[a] = p.mn[a] < mn[a] ? p.mn[a] : mn[a]; mx[a] = p.mx[a] > mx[a] ? p.mx[a] : mx[a]; }
    }
    void write_leaf(uint32_t pos, const Prim& p)
    {
        rt_bvh_node n = ref[p.leaf];
        n.num_primitives_axis &= 0xFFFF0000u;
        out[pos] = n;
    }
    void write_interior(uint32_t pos, const float mn[3], const float mx[3], uint32_t axis, uint32_t second)
    {
        rt_bvh_node n;
        memset(&n, 0, sizeof(n));
        n.bounds_min.x = mn[0]; n.bounds_min.y = mn[1]; n.bounds_min.z = mn[2];
        n.bounds_max.x = mx[0]; n.bounds_max.y = mx[1]; n.bounds_max.z = mx[2];
        n.offset = second;
        n.num_primitives_axis = axis;
        out[pos] = n;
    }

    // fn(slice, first, last) over K equal slices of [b, e), slice 0 on the calling thread.  Everything the slices reduce here (min / max of floats, counts) is
    // exact and order-free, so a result never depends on K.
    template <class F> static void slices(uint32_t b, uint32_t e, unsigned K, F fn)
    {
        const uint32_t n = e - b;
        if (K <= 1u || n < 2u * K) { fn(0u, b, e); return; }
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < K; ++k) pool.emplace_back(fn, k, b + (uint32_t)((uint64_t)n * k / K), b + (uint32_t)((uint64_t)n * (k + 1) / K));
        fn(0u, b, b + (uint32_t)((uint64_t)n / K));
        for (auto& th : pool) th.join();
    }

    // Splits prims[b, e) (n >= 2) in place; returns the first index of the upper part and the axis.  Cost of a candidate:
    // metric(lower box) * leaves below it + metric(upper box) * leaves above it (the greedy top-down SAH; interior boxes
    // are all that is paid for here -- every leaf is one reference leaf whichever tree it hangs in).
    // K: the threads this range may use for its passes (the top of a large tree: Builder::run); the split is the same for every K.
    uint32_t split(uint32_t b, uint32_t e, uint32_t& axis_out, std::vector<double>& scratch, std::vector<uint32_t>& order, unsigned K = 1)
    {
        const uint32_t n = e - b;
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (K > 1u)
        {
            std::vector<std::array<float, 6>> part(K, std::array<float, 6>{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY});
            slices(b, e, K, [&](unsigned k, uint32_t sb, uint32_t se)
            {
                std::array<float, 6> r = part[k];
                for (uint32_t i = sb; i < se; ++i)
                    for (int a = 0; a < 3; ++a) { r[a] = std::min(r[a], prims[i].c[a]); r[3 + a] = std::max(r[3 + a], prims[i].c[a]); }
                part[k] = r;
            });
            for (const auto& r : part) for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], r[a]); cmx[a] = std::max(cmx[a], r[3 + a]); }
        }
        else
            for (uint32_t i = b; i < e; ++i)
                for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], prims[i].c[a]); cmx[a] = std::max(cmx[a], prims[i].c[a]); }
        double best = INFINITY; int best_axis = -1; uint32_t best_at = 0;
        const float NINF = -INFINITY;
        if (n > 768u)
        {
            // binned: 256 (64 below 32 Ki leaves) bins over the centroid bounds of every axis
            const int NB = n > 32768u ? 256 : 64;
            struct Bin { float mn[3], mx[3]; uint32_t count; };
            Bin bins[256];
            for (int a = 0; a < 3; ++a)
            {
                if (!(cmx[a] > cmn[a])) continue;
                for (int j = 0; j < NB; ++j) { Bin& bn = bins[j]; for (int k = 0; k < 3; ++k) { bn.mn[k] = INFINITY; bn.mx[k] = NINF; } bn.count = 0; }
                const double scale = NB / ((double)cmx[a] - cmn[a]);
                auto bin_of = [&](const Prim& p) { int k = (int)(((double)p.c[a] - cmn[a]) * scale); return k < 0 ? 0 : (k >= NB ? NB - 1 : k); };
                if (K > 1u)
                {
                    // every slice bins into bins of its own; merged in slice order (min / max / sums of counts: the same bins whatever K is)
                    std::vector<Bin> part((size_t)K * NB);
                    slices(b, e, K, [&](unsigned k, uint32_t sb, uint32_t se)
                    {
                        Bin* mine = &part[(size_t)k * NB];
                        for (int j = 0; j < NB; ++j) { for (int q = 0; q < 3; ++q) { mine[j].mn[q] = INFINITY; mine[j].mx[q] = NINF; } mine[j].count = 0; }
                        for (uint32_t i = sb; i < se; ++i) { Bin& bn = mine[bin_of(prims[i])]; grow(bn.mn, bn.mx, prims[i]); ++bn.count; }
                    });
                    for (unsigned k = 0; k < K; ++k)
                        for (int j = 0; j < NB; ++j)
                        {
                            const Bin& src = part[(size_t)k * NB + j];
                            if (!src.count) continue;
                            Bin& bn = bins[j];
                            for (int q = 0; q < 3; ++q) { bn.mn[q] = std::min(bn.mn[q], src.mn[q]); bn.mx[q] = std::max(bn.mx[q], src.mx[q]); }
                            bn.count += src.count;
                        }
                }
                else
                    for (uint32_t i = b; i < e; ++i) { Bin& bn = bins[bin_of(prims[i])]; grow(bn.mn, bn.mx, prims[i]); ++bn.count; }
                scratch.assign(NB, 0.0);
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {NINF, NINF, NINF};
                uint32_t cnt = 0;
                for (int k = NB - 1; k > 0; --k)
                {
                    if (bins[k].count) { for (int q = 0; q < 3; ++q) { mn[q] = std::min(mn[q], bins[k].mn[q]); mx[q] = std::max(mx[q], bins[k].mx[q]); } cnt += bins[k].count; }
                    scratch[k] = cnt ? metric.of(mn, mx) * cnt : INFINITY;       // cost of the upper part [k, NB)
                }
                for (int q = 0; q < 3; ++q) { mn[q] = INFINITY; mx[q] = NINF; }
                cnt = 0;
                for (int k = 0; k < NB - 1; ++k)
                {
                    if (bins[k].count) { for (int q = 0; q < 3; ++q) { mn[q] = std::min(mn[q], bins[k].mn[q]); mx[q] = std::max(mx[q], bins[k].mx[q]); } cnt += bins[k].count; }
                    if (cnt == 0 || cnt == n) continue;
                    const double c = metric.of(mn, mx) * cnt + scratch[k + 1];
                    if (c < best) { best = c; best_axis = a; best_at = (uint32_t)k; }
                }
            }
            if (best_axis >= 0)
            {
                const int a = best_axis;
                const double scale = NB / ((double)cmx[a] - cmn[a]);
                auto bin_of = [&](const Prim& p) { int k = (int)(((double)p.c[a] - cmn[a]) * scale); return k < 0 ? 0 : (k >= NB ? NB - 1 : k); };
                Prim* mid = std::partition(&prims[b], &prims[b] + n, [&](const Prim& p) { return (uint32_t)bin_of(p) <= best_at; });
                axis_out = (uint32_t)a;
                return (uint32_t)(mid - &prims[0]);
            }
        }
        else
        {
            // full sweep: every border between two consecutive centroids, all three axes.  (Since build_small() takes every range of <= 768 leaves this branch is
            // its specification rather than its code path: build_small() visits the same candidates in the same order from orders sorted once.)
            typedef std::pair<float, uint32_t> Key;                        // (centroid, reference leaf): a total order
            std::vector<Key> keys(n), best_keys;
            std::vector<uint32_t> at(n);                                   // position in prims[] of the k-th key
            scratch.resize(n);
            order.resize(n);
            for (int a = 0; a < 3; ++a)
            {
                if (!(cmx[a] > cmn[a])) continue;
                for (uint32_t i = 0; i < n; ++i) keys[i] = Key(prims[b + i].c[a], i);
                std::sort(keys.begin(), keys.end(), [&](const Key& x, const Key& y)
                    { return x.first < y.first || (x.first == y.first && prims[b + x.second].leaf < prims[b + y.second].leaf); });
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {NINF, NINF, NINF};
                for (uint32_t i = n; i-- > 1;) { grow(mn, mx, prims[b + keys[i].second]); scratch[i] = metric.of(mn, mx) * (n - i); }
                for (int q = 0; q < 3; ++q) { mn[q] = INFINITY; mx[q] = NINF; }
                bool improved = false;
                for (uint32_t i = 1; i < n; ++i)
                {
                    grow(mn, mx, prims[b + keys[i - 1].second]);
                    const double c = metric.of(mn, mx) * i + scratch[i];
                    if (c < best) { best = c; best_axis = a; best_at = i; improved = true; }
                }
                if (improved) for (uint32_t i = 0; i < n; ++i) order[i] = keys[i].second;
            }
            if (best_axis >= 0)
            {
                // apply: the `best_at` lowest in that axis' order go first
                std::vector<Prim> tmp(n);
                for (uint32_t i = 0; i < n; ++i) tmp[i] = prims[b + order[i]];
                std::copy(tmp.begin(), tmp.end(), prims.begin() + b);
                axis_out = (uint32_t)best_axis;
                return b + best_at;
            }
        }
        // all centroids coincide (or the bins could not separate them): halve by reference order
        std::sort(&prims[b], &prims[b] + n, [](const Prim& x, const Prim& y) { return x.leaf < y.leaf; });
        int a = 0;
        for (int q = 1; q < 3; ++q) if (cmx[q] - cmn[q] > cmx[a] - cmn[a]) a = q;
        axis_out = (uint32_t)a;
        return b + n / 2;
    }

    // One node: the box of prims[b, e) (n >= 2), its split, its record at pos.  Returns the first index of the upper part.
    // a subtree over n leaves takes 2 n - 1 records: first child at pos + 1, second at pos + 2 * (leaves of the first)
    uint32_t step(uint32_t b, uint32_t e, uint32_t pos, std::vector<double>& scratch, std::vector<uint32_t>& order, unsigned K = 1)
    {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (K > 1u)
        {
            std::vector<std::array<float, 6>> part(K, std::array<float, 6>{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY});
            slices(b, e, K, [&](unsigned k, uint32_t sb, uint32_t se)
            {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (uint32_t i = sb; i < se; ++i) grow(lo, hi, prims[i]);
                part[k] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
            });
            for (const auto& r : part) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], r[a]); mx[a] = std::max(mx[a], r[3 + a]); }
        }
        else
            for (uint32_t i = b; i < e; ++i) grow(mn, mx, prims[i]);
        uint32_t axis = 0;
        const uint32_t mid = split(b, e, axis, scratch, order, K);
        write_interior(pos, mn, mx, axis, pos + 2u * (mid - b));
        return mid;
    }

    // A subtree of at most 768 leaves -- where split() sweeps every border on all three axes -- built from orders sorted ONCE: split() sorts the range three times at
    // every node, this keeps the three orders of the subtree's leaves (by centroid, then reference leaf: the total order split() sorts by) and partitions them
    // stably into the two parts at every node, so each node's sweeps see exactly the sequences split() would have sorted.  The same candidates in the same order
    // with the same arithmetic: the same tree (tools/own_bvh_bench.cpp compares; 55 % of a build's one-thread time were those sorts).  prims[] is not permuted:
    // which leaf ends where follows from set membership alone.
    struct Small
    {
        uint16_t ord[3][768];              // the subtree's leaves (indices relative to b) in each axis' order; a node owns the same segment [s, s + m) of all three
        uint16_t tmp[768];
        uint8_t lower[768];
        double suffix[768];
    };
    void small_node(Small& S, uint32_t b, uint32_t s, uint32_t m, uint32_t pos)
    {
        if (m == 1) { write_leaf(pos, prims[b + S.ord[0][s]]); return; }
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < m; ++i) grow(mn, mx, prims[b + S.ord[0][s + i]]);
        const float NINF = -INFINITY;
        double best = INFINITY; int best_axis = -1; uint32_t best_at = 0;
        float cmn[3], cmx[3];
        for (int a = 0; a < 3; ++a) { cmn[a] = prims[b + S.ord[a][s]].c[a]; cmx[a] = prims[b + S.ord[a][s + m - 1u]].c[a]; }
        for (int a = 0; a < 3; ++a)
        {
            if (!(cmx[a] > cmn[a])) continue;
            const uint16_t* o = &S.ord[a][s];
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {NINF, NINF, NINF};
            for (uint32_t i = m; i-- > 1;) { grow(lo, hi, prims[b + o[i]]); S.suffix[i] = metric.of(lo, hi) * (m - i); }
            for (int q = 0; q < 3; ++q) { lo[q] = INFINITY; hi[q] = NINF; }
            for (uint32_t i = 1; i < m; ++i)
            {
                grow(lo, hi, prims[b + o[i - 1]]);
                const double c = metric.of(lo, hi) * i + S.suffix[i];
                if (c < best) { best = c; best_axis = a; best_at = i; }
            }
        }
        uint32_t nl, axis;
        if (best_axis >= 0)
        {
            nl = best_at; axis = (uint32_t)best_axis;
            for (uint32_t i = 0; i < m; ++i) S.lower[S.ord[best_axis][s + i]] = i < nl ? 1 : 0;
        }
        else
        {
            // all centroids coincide: halve by reference order (split()'s last resort)
            for (uint32_t i = 0; i < m; ++i) S.tmp[i] = S.ord[0][s + i];
            std::sort(S.tmp, S.tmp + m, [&](uint16_t x, uint16_t y) { return prims[b + x].leaf < prims[b + y].leaf; });
            nl = m / 2u;
            for (uint32_t i = 0; i < m; ++i) S.lower[S.tmp[i]] = i < nl ? 1 : 0;
            int a = 0;
            for (int q = 1; q < 3; ++q) if (cmx[q] - cmn[q] > cmx[a] - cmn[a]) a = q;
            axis = (uint32_t)a;
        }
        for (int a = 0; a < 3; ++a)
        {
            uint16_t* o = &S.ord[a][s];
            uint32_t at = 0, up = 0;
            for (uint32_t i = 0; i < m; ++i) { if (S.lower[o[i]]) o[at++] = o[i]; else S.tmp[up++] = o[i]; }      // (at <= i: nothing unread is overwritten)
            for (uint32_t i = 0; i < up; ++i) o[at + i] = S.tmp[i];
        }
        write_interior(pos, mn, mx, axis, pos + 2u * nl);
        small_node(S, b, s, nl, pos + 1u);
        small_node(S, b, s + nl, m - nl, pos + 2u * nl);
    }
    void build_small(uint32_t b, uint32_t e, uint32_t pos)
    {
        const uint32_t n = e - b;
        std::unique_ptr<Small> S(new Small);
        for (int a = 0; a < 3; ++a)
        {
            for (uint32_t i = 0; i < n; ++i) S->ord[a][i] = (uint16_t)i;
            std::sort(S->ord[a], S->ord[a] + n, [&](uint16_t x, uint16_t y)
                { const Prim &p = prims[b + x], &q = prims[b + y]; return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.leaf < q.leaf); });
        }
        small_node(*S, b, 0u, n, pos);
    }

    // the whole subtree over prims[b, e) at pos, on the calling thread
    void build(uint32_t b, uint32_t e, uint32_t pos, std::vector<double>& scratch, std::vector<uint32_t>& order)
    {
        for (;;)
        {
            const uint32_t n = e - b;
            if (n > 4096u && cancelled()) return;                           // (the result is dropped by build())
            if (n == 1) { write_leaf(pos, prims[b]); return; }
            if (n <= 768u) { build_small(b, e, pos); return; }             // (split() would take its sweep branch for this node and every node below it)
            const uint32_t mid = step(b, e, pos, scratch, order);
            const uint32_t nl = mid - b;
            // recurse into the SMALLER part, iterate on the larger: the positions of both are known (pos + 1 and pos + 2 nl), so the
            // order does not matter and the recursion is at most log2(n) deep however lopsided the splits are
            if (nl <= n - nl) { build(b, mid, pos + 1u, scratch, order); b = mid; pos = pos + 2u * nl; }
            else { build(mid, e, pos + 2u * nl, scratch, order); e = mid; pos = pos + 1u; }
        }
    }

    // The tree, on n_threads threads.  A queue of ranges, the largest taken first: a range above the grain is split ONCE (its passes over the leaves on as many
    // threads as its share of all leaves is of the pool: the root's on all of them, its children's on half each ...) and hands its two parts back to the queue; a
    // range within the grain is built to the bottom by the thread that took it.  Which thread does what changes nothing in the result: a split depends on the SET
    // of leaves in its range only (bins are min / max / counts, the sweep sorts by a total order), and every record's position follows from the counts.
    // (Round 6: the top of the tree -- nine levels for 8.7 M leaves -- used to be built by one thread before the pool started: 1.8 s of which 1.5 were that.)
    void run()
    {
        const uint32_t np = (uint32_t)prims.size();
        std::mutex mu;
        std::condition_variable cv;
        std::vector<Task> queue{Task{0u, np, 0u}};
        unsigned active = 0;
        auto worker = [&]()
        {
            std::vector<double> scratch;
            std::vector<uint32_t> order;
            for (;;)
            {
                Task t;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !queue.empty() || active == 0u || cancelled(); });
                    if (cancelled() || queue.empty()) { cv.notify_all(); return; }      // (empty and nobody active: done)
                    size_t at = 0;
                    for (size_t i = 1; i < queue.size(); ++i) if (queue[i].e - queue[i].b > queue[at].e - queue[at].b) at = i;
                    t = queue[at];
                    queue[at] = queue.back();
                    queue.pop_back();
                    ++active;
                }
                const uint32_t n = t.e - t.b;
                Task kids[2];
                unsigned n_kids = 0;
                if (n <= grain || n < 2u) build(t.b, t.e, t.pos, scratch, order);
                else
                {
                    const unsigned K = (unsigned)std::max<uint64_t>(1u, std::min<uint64_t>(n_threads, ((uint64_t)n_threads * n + np / 2u) / np));
                    const uint32_t mid = step(t.b, t.e, t.pos, scratch, order, n >= 65536u ? K : 1u);
                    kids[0] = Task{t.b, mid, t.pos + 1u};
                    kids[1] = Task{mid, t.e, t.pos + 2u * (mid - t.b)};
                    n_kids = 2;
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    for (unsigned k = 0; k < n_kids; ++k) queue.push_back(kids[k]);
                    --active;
                }
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
};

// nodes[nn]: the reference's LinearBVHNode[] (validated by the caller: build_wide_bvh's pass 0 has the same requirements).
// false: nothing to do (leaf root) or the array is not a tree.
// cancel (optional): raised by another thread -- the build gives up at its next check and returns false (rt_scene_upload: a candidate built on the device won already)
// threads: 0 = the host's (at most 32); the tree does not depend on it
// phases (optional, tools/own_bvh_bench.cpp): seconds of { collecting the leaves, allocating the output, the pool }
inline bool build(const rt_bvh_node* nodes, uint32_t nn, const Metric& metric, std::vector<rt_bvh_node>& out, const std::atomic<bool>* cancel = nullptr, unsigned threads = 0,
    double* phases = nullptr)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    out.clear();
    if (nn == 0 || (nodes[0].num_primitives_axis >> 16) != 0) return false;
    Builder B;
    B.ref = nodes;
    B.metric = metric;
    B.cancel = cancel;
    B.n_threads = threads ? threads : (unsigned)std::min<size_t>(std::max(1u, std::min(std::thread::hardware_concurrency(), 32u)), (size_t)nn / 8192u + 1u);   // (a small tree: few threads)
    {
        // the walk only lists the leaves (and is what says "not a tree"); their boxes are read by the pool's threads afterwards
        std::vector<uint32_t> leaves;
        leaves.reserve((size_t)nn / 2u + 1u);
        std::vector<uint32_t> todo{0u};
        size_t seen = 0;
        while (!todo.empty())
        {
            const uint32_t i = todo.back();
            todo.pop_back();
            if (++seen > nn) return false;
            const rt_bvh_node& n = nodes[i];
            if ((n.num_primitives_axis >> 16) != 0) { leaves.push_back(i); continue; }
            if (i + 1 >= nn || n.offset >= nn || n.offset <= i + 1) return false;
            todo.push_back(n.offset);
            todo.push_back(i + 1);
        }
        // (the output array -- 2 n - 1 records, value-initialised by one thread whatever is done about it -- is allocated beside the leaves' array and its filling)
        struct Beside { std::thread th; ~Beside() { if (th.joinable()) th.join(); } } beside;
        if (leaves.size() >= 65536u) beside.th = std::thread([&B, n = leaves.size()]() { B.out.resize(2u * n - 1u); });
        B.prims.resize(leaves.size());
        std::atomic<bool> finite{true};
        Builder::slices(0u, (uint32_t)leaves.size(), leaves.size() >= 65536u ? B.n_threads : 1u, [&](unsigned, uint32_t sb, uint32_t se)
        {
            for (uint32_t k = sb; k < se; ++k)
            {
                const rt_bvh_node& n = nodes[leaves[k]];
                Prim p;
                p.mn[0] = n.bounds_min.x; p.mn[1] = n.bounds_min.y; p.mn[2] = n.bounds_min.z;
                p.mx[0] = n.bounds_max.x; p.mx[1] = n.bounds_max.y; p.mx[2] = n.bounds_max.z;
                for (int a = 0; a < 3; ++a)
                {
                    if (!std::isfinite(p.mn[a]) || !std::isfinite(p.mx[a])) finite.store(false, std::memory_order_relaxed);
                    p.c[a] = 0.5f * p.mn[a] + 0.5f * p.mx[a];
                }
                p.leaf = leaves[k];
                B.prims[k] = p;
            }
        });
        if (!finite.load()) return false;
    }
    const uint32_t np = (uint32_t)B.prims.size();
    if (np < 2) return false;
    const double t_collected = since();
    if (B.out.size() != (size_t)2 * np - 1) B.out.resize((size_t)2 * np - 1);
    const double t_allocated = since();
    B.run();
    if (phases) { phases[0] = t_collected; phases[1] = t_allocated - t_collected; phases[2] = since() - t_allocated; }
    if (B.cancelled()) return false;
    out.swap(B.out);
    return true;
}
} // namespace ownbvh

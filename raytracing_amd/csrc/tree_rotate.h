// tree_rotate.h -- local search on a binary tree over the reference's leaves with a MEASURED objective (host code; round 4, analysis /
// opt-in: RT_CTX_OPT_ADAPTIVE_FOLD bit 3).
//
// For any-hit (shadow) rays every binary tree over the reference's leaves gives the reference's verdict (own_bvh.h), so the tree may be
// whatever is cheapest for the rays that are actually traced.  own_bvh.h builds one from a geometric model of those rays (projected area
// along the lights); once a probe frame exists (rt_hip.hip: FoldAdapt) the model can be replaced by counting: the cost of an interior node
// is the number of probe rays whose segment crosses its box -- that is how often a walk visits it -- and the tree's cost the sum over its
// interior nodes.  This file lowers that sum by TREE ROTATIONS (Kensler 2008, "Tree rotations for improving bounding volume hierarchies",
// with the surface area replaced by the crossing count): at a node n = (L, R) with L = (L1, L2), exchanging R with L1 or L2 changes one box
// only -- L's -- and exchanging a grandchild under L with one under R changes two; the rays that can cross a new box are among those
// crossing n, which each node keeps as a list, so a candidate costs one box test per listed ray.  Leaves, their boxes and their triangles
// are never touched; the boxes of all other nodes stay exact unions by construction.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>
#include "rt_types.h"

namespace treerot
{
struct Ray { float o[3], inv[3], t_max; };

struct Tree
{
    // pointer form: node i has children kid0[i], kid1[i] (NONE for a leaf) and box mn / mx
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> kid0, kid1;                           // (node ids are the input's: a leaf i is nodes[i])
    std::vector<float> mn, mx;                                  // 3 floats per node
    std::vector<std::vector<uint32_t>> rays;                    // the probe rays crossing each interior node's box (empty for leaves)
    uint32_t root = 0;
    bool leaf(uint32_t i) const { return kid0[i] == NONE; }
};

inline bool crosses(const Ray& r, const float* mn, const float* mx)
{
    float t0 = 0.0f, t1 = r.t_max;
    for (int a = 0; a < 3; ++a)
    {
        const float ta = (mn[a] - r.o[a]) * r.inv[a], tb = (mx[a] - r.o[a]) * r.inv[a];
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
    }
    return t0 <= t1;
}

// nodes: a binary tree in the reference's linear layout (validated by the caller).  Returns the number of rotations made; `out` is the
// rotated tree in the same layout (split axis of an interior node = the axis along which its children's box centres are farthest apart).
// cost[0] / cost[1] = sum of crossing counts over interior nodes before / after, per ray.
inline uint32_t rotate(const rt_bvh_node* nodes, uint32_t nn, const float* origins_tmax, const float* directions, size_t n_rays, int max_passes,
    std::vector<rt_bvh_node>& out, double cost[2], const std::atomic<bool>* cancel = nullptr,
    int moves = 3 /* bit 0: child <-> grandchild, bit 1: grandchild <-> grandchild */,
    double min_gain = 0.03 /* a move must save more than this share of the crossings of the node it is made at: a search that takes every small gain
                              locks itself in (a probe twice as large changes nothing, so it is the greed, not the sample) -- 0 / 0.03 / 0.1: 5.01 /
                              4.60 / 4.64 steps per unseen shadow ray on a 300 K-triangle scene with both move kinds; 11.43 at 0.03 against 11.98 with
                              the first kind alone on the 2.8 M one (tools/fold_weight_study.py --tree) */,
    double* phases = nullptr /* seconds of { pointer form, the rays' lists, the passes, back to the linear layout } */,
    unsigned threads = 0 /* 0 = the host's, at most 16; the rotated tree does not depend on it */, size_t grain = 256 /* a child with a list this long is another thread's */)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&, last = 0.0](int k) mutable { const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); if (phases) phases[k] = t - last; last = t; };
    out.clear();
    cost[0] = cost[1] = 0.0;
    if (nn == 0 || n_rays == 0) return 0;
    Tree t;
    t.kid0.assign(nn, Tree::NONE); t.kid1.assign(nn, Tree::NONE);
    t.mn.resize((size_t)nn * 3); t.mx.resize((size_t)nn * 3);
    t.rays.resize(nn);
    // the caller may be an exported test hook (rt_debug_rotate_tree): every node but the root must be the child of exactly one interior
    // node -- a shared child (a DAG) would be walked once per path to it and only fail the size check at the very end
    const unsigned n_threads = threads ? threads : (unsigned)std::min<size_t>(std::max(1u, std::min(std::thread::hardware_concurrency(), 16u)), n_rays / 512u + 1u);   // (a small probe: few threads)
    auto is_cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
    // fn(k) for k = 0 .. K - 1, k = 0 on the calling thread
    auto on_threads = [&](unsigned K, auto fn)
    {
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < K; ++k) pool.emplace_back(fn, k);
        fn(0u);
        for (auto& th : pool) th.join();
    };
    {
        std::unique_ptr<std::atomic<uint8_t>[]> referenced(new std::atomic<uint8_t>[nn]);
        std::atomic<bool> bad{false};
        const unsigned K = nn >= 65536u ? n_threads : 1u;
        on_threads(K, [&](unsigned k) { for (uint32_t i = (uint32_t)((uint64_t)nn * k / K), e = (uint32_t)((uint64_t)nn * (k + 1) / K); i < e; ++i) referenced[i].store(0, std::memory_order_relaxed); });
        on_threads(K, [&](unsigned k)
        {
            for (uint32_t i = (uint32_t)((uint64_t)nn * k / K), e = (uint32_t)((uint64_t)nn * (k + 1) / K); i < e; ++i)
            {
                const rt_bvh_node& n = nodes[i];
                t.mn[3 * (size_t)i] = n.bounds_min.x; t.mn[3 * (size_t)i + 1] = n.bounds_min.y; t.mn[3 * (size_t)i + 2] = n.bounds_min.z;
                t.mx[3 * (size_t)i] = n.bounds_max.x; t.mx[3 * (size_t)i + 1] = n.bounds_max.y; t.mx[3 * (size_t)i + 2] = n.bounds_max.z;
                if ((n.num_primitives_axis >> 16) == 0)
                {
                    if (i + 1 >= nn || n.offset >= nn || n.offset <= i + 1) { bad.store(true); return; }
                    if (referenced[i + 1].fetch_add(1, std::memory_order_relaxed) != 0 || referenced[n.offset].fetch_add(1, std::memory_order_relaxed) != 0) { bad.store(true); return; }
                    t.kid0[i] = i + 1; t.kid1[i] = n.offset;
                }
            }
        });
        if (bad.load()) return 0;
    }
    lap(0);
    std::vector<Ray> rays(n_rays);
    for (size_t r = 0; r < n_rays; ++r)
    {
        Ray& q = rays[r];
        for (int a = 0; a < 3; ++a) { q.o[a] = origins_tmax[4 * r + a]; q.inv[a] = 1.0f / directions[4 * r + a]; }
        q.t_max = origins_tmax[4 * r + 3];
    }
    // the lists: every ray walks the tree once -- the rays in equal slices on the pool's threads, each noting (node, ray) as it goes; the notes become the
    // lists slice after slice, so every list is in ascending ray order whatever the number of threads
    {
        const unsigned K = (unsigned)std::min<size_t>(n_threads, n_rays / 1024u + 1u);
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> notes(K);
        auto walk = [&](unsigned k)
        {
            std::vector<uint32_t> stack;
            auto& mine = notes[k];
            for (size_t r = n_rays * k / K, r1 = n_rays * (k + 1) / K; r < r1; ++r)
            {
                stack.assign(1, 0u);
                while (!stack.empty())
                {
                    const uint32_t i = stack.back();
                    stack.pop_back();
                    if (!crosses(rays[r], &t.mn[3 * (size_t)i], &t.mx[3 * (size_t)i])) continue;
                    if (t.leaf(i)) continue;
                    mine.emplace_back(i, (uint32_t)r);
                    stack.push_back(t.kid1[i]);
                    stack.push_back(t.kid0[i]);
                }
                if ((r & 1023u) == 0u && is_cancelled()) return;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < K; ++k) pool.emplace_back(walk, k);
        walk(0u);
        for (auto& th : pool) th.join();
        if (is_cancelled()) return 0;
        std::vector<uint32_t> count(nn, 0u);
        for (const auto& part : notes) for (const auto& nr : part) ++count[nr.first];
        for (uint32_t i = 0; i < nn; ++i) if (count[i]) t.rays[i].reserve(count[i]);
        for (const auto& part : notes) for (const auto& nr : part) t.rays[nr.first].push_back(nr.second);
    }
    auto total = [&]() { double s = 0.0; for (uint32_t i = 0; i < nn; ++i) s += (double)t.rays[i].size(); return s / (double)n_rays; };
    cost[0] = total();
    lap(1);
    // rotations, top-down over the nodes rays reach (a node without rays has nothing to gain), repeated until a pass changes little.
    // What a visit of node n reads and writes lies in n's subtree (its children's and grandchildren's links, its children's boxes and lists), and it must come
    // after the visits of n's ancestors in the same pass -- nothing else orders two visits, so disjoint subtrees are visited by different threads: a visited node
    // hands each child whose list is at least `grain` rays long to a shared queue and walks the smaller ones itself, breadth first.  The result is the serial
    // pass's whatever the threads do (round 6: the passes were 0.7 of an adaptation's 1.2 s on the headline scene, on one thread).
    struct Scratch { std::vector<uint32_t> keep, keep2, best_list, best_list2, order; };
    auto visit = [&](uint32_t n, Scratch& sc) -> uint32_t
    {
        std::vector<uint32_t>&keep = sc.keep, &keep2 = sc.keep2, &best_list = sc.best_list, &best_list2 = sc.best_list2;
        // candidates: (a) exchange one child of n with one grandchild under the OTHER child -- one box changes; (b) exchange a grandchild
        // under one child with a grandchild under the other -- both children's boxes change
        long best_gain = (long)(min_gain * (double)t.rays[n].size()); int best_kind = -1, best_side = -1, best_g = -1, best_h = -1;
        for (int side = 0; side < 2 && (moves & 1); ++side)
        {
            const uint32_t c = side ? t.kid1[n] : t.kid0[n];             // the child that is opened
            const uint32_t other = side ? t.kid0[n] : t.kid1[n];          // the child that moves down
            if (t.leaf(c)) continue;
            for (int g = 0; g < 2; ++g)
            {
                const uint32_t stay = g ? t.kid0[c] : t.kid1[c];          // (the grandchild that moves up is the other one)
                float bmn[3], bmx[3];
                for (int a = 0; a < 3; ++a)
                {
                    bmn[a] = std::min(t.mn[3 * (size_t)other + a], t.mn[3 * (size_t)stay + a]);
                    bmx[a] = std::max(t.mx[3 * (size_t)other + a], t.mx[3 * (size_t)stay + a]);
                }
                keep.clear();
                for (uint32_t r : t.rays[n]) if (crosses(rays[r], bmn, bmx)) keep.push_back(r);
                const long gain = (long)t.rays[c].size() - (long)keep.size();   // c's box becomes (other + stay)'s
                if (gain > best_gain) { best_gain = gain; best_kind = 0; best_side = side; best_g = g; best_list = keep; }
            }
        }
        const uint32_t L = t.kid0[n], R = t.kid1[n];
        if ((moves & 2) && !t.leaf(L) && !t.leaf(R))
            for (int g = 0; g < 2; ++g)
                for (int h = 0; h < 2; ++h)
                {
                    // L = (lg, lo), R = (rh, ro)  ->  L = (rh, lo), R = (lg, ro); (g, h) and (1 - g, 1 - h) give the same pair of sets: h <= g suffices
                    if (h > g) continue;
                    const uint32_t lg = g ? t.kid1[L] : t.kid0[L], lo = g ? t.kid0[L] : t.kid1[L];
                    const uint32_t rh = h ? t.kid1[R] : t.kid0[R], ro = h ? t.kid0[R] : t.kid1[R];
                    float amn[3], amx[3], bmn[3], bmx[3];
                    for (int a = 0; a < 3; ++a)
                    {
                        amn[a] = std::min(t.mn[3 * (size_t)rh + a], t.mn[3 * (size_t)lo + a]); amx[a] = std::max(t.mx[3 * (size_t)rh + a], t.mx[3 * (size_t)lo + a]);
                        bmn[a] = std::min(t.mn[3 * (size_t)lg + a], t.mn[3 * (size_t)ro + a]); bmx[a] = std::max(t.mx[3 * (size_t)lg + a], t.mx[3 * (size_t)ro + a]);
                    }
                    keep.clear(); keep2.clear();
                    for (uint32_t r : t.rays[n])
                    {
                        if (crosses(rays[r], amn, amx)) keep.push_back(r);
                        if (crosses(rays[r], bmn, bmx)) keep2.push_back(r);
                    }
                    const long gain = (long)t.rays[L].size() + (long)t.rays[R].size() - (long)keep.size() - (long)keep2.size();
                    if (gain > best_gain) { best_gain = gain; best_kind = 1; best_g = g; best_h = h; best_list = keep; best_list2 = keep2; }
                }
        if (best_kind == 0)
        {
            const uint32_t c = best_side ? t.kid1[n] : t.kid0[n];
            const uint32_t other = best_side ? t.kid0[n] : t.kid1[n];
            const uint32_t up = best_g ? t.kid1[c] : t.kid0[c];
            const uint32_t stay = best_g ? t.kid0[c] : t.kid1[c];
            // n = (c, other), c = (up, stay)  ->  n = (c, up), c = (other, stay)
            (best_side ? t.kid0[n] : t.kid1[n]) = up;
            t.kid0[c] = other; t.kid1[c] = stay;
            for (int a = 0; a < 3; ++a)
            {
                t.mn[3 * (size_t)c + a] = std::min(t.mn[3 * (size_t)other + a], t.mn[3 * (size_t)stay + a]);
                t.mx[3 * (size_t)c + a] = std::max(t.mx[3 * (size_t)other + a], t.mx[3 * (size_t)stay + a]);
            }
            t.rays[c].swap(best_list);
            return 1u;
        }
        if (best_kind == 1)
        {
            const uint32_t lg = best_g ? t.kid1[L] : t.kid0[L], lo = best_g ? t.kid0[L] : t.kid1[L];
            const uint32_t rh = best_h ? t.kid1[R] : t.kid0[R], ro = best_h ? t.kid0[R] : t.kid1[R];
            t.kid0[L] = rh; t.kid1[L] = lo;
            t.kid0[R] = lg; t.kid1[R] = ro;
            for (int a = 0; a < 3; ++a)
            {
                t.mn[3 * (size_t)L + a] = std::min(t.mn[3 * (size_t)rh + a], t.mn[3 * (size_t)lo + a]); t.mx[3 * (size_t)L + a] = std::max(t.mx[3 * (size_t)rh + a], t.mx[3 * (size_t)lo + a]);
                t.mn[3 * (size_t)R + a] = std::min(t.mn[3 * (size_t)lg + a], t.mn[3 * (size_t)ro + a]); t.mx[3 * (size_t)R + a] = std::max(t.mx[3 * (size_t)lg + a], t.mx[3 * (size_t)ro + a]);
            }
            t.rays[L].swap(best_list);
            t.rays[R].swap(best_list2);
            return 1u;
        }
        return 0u;
    };
    uint32_t rotations = 0;
    for (int pass = 0; pass < max_passes; ++pass)
    {
        std::atomic<uint32_t> made{0};
        std::mutex mu;
        std::condition_variable cv;
        typedef std::pair<size_t, uint32_t> Job;                               // (rays in the list when it was handed over, subtree root): the longest list first
        std::priority_queue<Job> queue;                                        // subtree roots whose ancestors have been visited in this pass
        queue.push(Job(t.rays[t.root].size(), t.root));
        unsigned active = 0;
        auto worker = [&]()
        {
            Scratch sc;
            std::vector<uint32_t> hand_over;
            for (;;)
            {
                uint32_t top;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !queue.empty() || active == 0u || is_cancelled(); });
                    if (is_cancelled() || queue.empty()) { cv.notify_all(); return; }
                    top = queue.top().second;
                    queue.pop();
                    ++active;
                }
                uint32_t made_here = 0;
                hand_over.clear();
                sc.order.assign(1, top);
                for (size_t head = 0; head < sc.order.size(); ++head)
                {
                    const uint32_t n = sc.order[head];
                    if (t.leaf(n) || t.rays[n].empty()) continue;
                    if ((head & 1023u) == 0u && is_cancelled()) break;
                    made_here += visit(n, sc);
                    for (uint32_t c : {t.kid0[n], t.kid1[n]})
                    {
                        if (t.leaf(c) || t.rays[c].empty()) continue;
                        if (n_threads > 1u && t.rays[c].size() >= grain) hand_over.push_back(c); else sc.order.push_back(c);
                    }
                    if (!hand_over.empty())
                    {
                        { std::lock_guard<std::mutex> lk(mu); for (uint32_t c : hand_over) queue.push(Job(t.rays[c].size(), c)); }
                        cv.notify_all();
                        hand_over.clear();
                    }
                }
                made.fetch_add(made_here);
                { std::lock_guard<std::mutex> lk(mu); --active; }
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < n_threads; ++k) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
        if (is_cancelled()) return 0;
        rotations += made.load();
        if (made.load() == 0) break;
    }
    cost[1] = total();
    lap(2);
    // back to the linear layout: depth first, first child at i + 1.  The top of the tree (to depth 10) on this thread, the subtrees below it on the pool: sizes
    // bottom-up first, then every node's position follows from its parent's and its sibling's size
    out.resize(nn);
    std::vector<uint32_t> size(nn, 1u);
    struct Item { uint32_t node, pos; };
    std::vector<uint32_t> tops;                                            // interior nodes above the cut, parents before children
    std::vector<Item> cuts;                                                // the subtrees' roots (pos: filled in below)
    {
        std::vector<std::pair<uint32_t, uint32_t>> level{{t.root, 0u}};
        for (size_t head = 0; head < level.size(); ++head)
        {
            const uint32_t i = level[head].first, depth = level[head].second;
            if (level.size() > (size_t)nn) { out.clear(); return 0; }      // (a cycle among the top nodes)
            if (t.leaf(i) || depth >= 10u) { cuts.push_back({i, 0u}); continue; }
            tops.push_back(i);
            level.push_back({t.kid0[i], depth + 1u});
            level.push_back({t.kid1[i], depth + 1u});
        }
    }
    std::atomic<bool> broken{false};
    std::atomic<size_t> next_cut{0};
    const unsigned K_layout = nn >= 65536u ? n_threads : 1u;
    on_threads(K_layout, [&](unsigned)
    {
        // subtree sizes, children before parents: an explicit post-order
        std::vector<std::pair<uint32_t, int>> st;
        for (size_t c; (c = next_cut.fetch_add(1)) < cuts.size();)
        {
            st.assign(1, {cuts[c].node, 0});
            size_t steps = 0;
            while (!st.empty())
            {
                if (++steps > 3u * (size_t)nn + 3u) { broken.store(true); return; }
                auto& top = st.back();
                const uint32_t i = top.first;
                if (t.leaf(i)) { st.pop_back(); continue; }
                if (top.second == 0) { top.second = 1; st.push_back({t.kid0[i], 0}); }
                else if (top.second == 1) { top.second = 2; st.push_back({t.kid1[i], 0}); }
                else { size[i] = 1u + size[t.kid0[i]] + size[t.kid1[i]]; st.pop_back(); }
            }
        }
    });
    if (broken.load()) { out.clear(); return 0; }
    for (size_t k = tops.size(); k-- > 0;) size[tops[k]] = 1u + size[t.kid0[tops[k]]] + size[t.kid1[tops[k]]];
    if (size[t.root] != nn) { out.clear(); return 0; }
    auto emit = [&](uint32_t i, uint32_t pos)                              // one interior node's record; returns its second child's position
    {
        rt_bvh_node n;
        memset(&n, 0, sizeof(n));
        n.bounds_min.x = t.mn[3 * (size_t)i]; n.bounds_min.y = t.mn[3 * (size_t)i + 1]; n.bounds_min.z = t.mn[3 * (size_t)i + 2];
        n.bounds_max.x = t.mx[3 * (size_t)i]; n.bounds_max.y = t.mx[3 * (size_t)i + 1]; n.bounds_max.z = t.mx[3 * (size_t)i + 2];
        const uint32_t a = t.kid0[i], b = t.kid1[i];
        int axis = 0; float far = -1.0f;
        for (int k = 0; k < 3; ++k)
        {
            const float ca = t.mn[3 * (size_t)a + k] + t.mx[3 * (size_t)a + k], cb = t.mn[3 * (size_t)b + k] + t.mx[3 * (size_t)b + k];
            const float d = fabsf(ca - cb);
            if (d > far) { far = d; axis = k; }
        }
        n.offset = pos + 1u + size[a];
        n.num_primitives_axis = (uint32_t)axis;
        out[pos] = n;
        return n.offset;
    };
    {
        // positions of the top nodes and of the cuts, in the order they were found (a parent's is known before its children's)
        std::vector<uint32_t> pos_of_level{0u};
        size_t k_top = 0, k_cut = 0;
        std::vector<std::pair<uint32_t, uint32_t>> level{{t.root, 0u}};
        for (size_t head = 0; head < level.size(); ++head)
        {
            const uint32_t i = level[head].first, depth = level[head].second, pos = pos_of_level[head];
            if (t.leaf(i) || depth >= 10u) { cuts[k_cut++].pos = pos; continue; }
            ++k_top;
            const uint32_t second = emit(i, pos);
            level.push_back({t.kid0[i], depth + 1u}); pos_of_level.push_back(pos + 1u);
            level.push_back({t.kid1[i], depth + 1u}); pos_of_level.push_back(second);
        }
        (void)k_top;
    }
    next_cut.store(0);
    on_threads(K_layout, [&](unsigned)
    {
        std::vector<Item> st;
        for (size_t c; (c = next_cut.fetch_add(1)) < cuts.size();)
        {
            st.assign(1, cuts[c]);
            while (!st.empty())
            {
                const Item it = st.back();
                st.pop_back();
                const uint32_t i = it.node;
                if (t.leaf(i)) { rt_bvh_node n = nodes[i]; n.num_primitives_axis &= 0xFFFF0000u; out[it.pos] = n; continue; }
                const uint32_t second = emit(i, it.pos);
                st.push_back({t.kid1[i], second});
                st.push_back({t.kid0[i], it.pos + 1u});
            }
        }
    });
    lap(3);
    return rotations;
}
} // namespace treerot

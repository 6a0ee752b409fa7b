// treelet_order.h -- the order of the child-pair records of the exact BVH2 (rt_scene_upload; host code).
//
// Record order = cache-friendly "treelet" layout: a breadth-first cluster of up to `treelet` interior nodes is stored contiguously (7 records = 448 B, i.e. the
// next three levels below a node share a few cache lines), then the clusters hanging off it, depth first.  A pure permutation of records: traversal decisions and
// results do not depend on it.  (treelet = 1 gives the reference's depth-first order.)
//
// On several threads (round 6: 0.11 s of the 10 M-triangle scene's upload on one): this thread walks the clusters down to cluster depth 4 and notes, in the order it
// meets them, its own clusters' nodes and the roots of the cluster subtrees below; the pool orders those subtrees (the same walk, into lists of their own); the lists,
// spliced in where their roots were met, are the one-thread order.
#pragma once
#include <stdint.h>
#include <atomic>
#include <thread>
#include <vector>
#include "rt_types.h"

namespace treelet
{
// interior_index[n] = position of interior node n's record (0xFFFFFFFF for leaves); returns 0 on success, 1: a child index outside the array (or not after its
// parent), 2: not a tree (a cycle or a shared child).  threads: 0 = the host's (at most 16); the order does not depend on it.
inline int order(const rt_bvh_node* nodes, uint32_t nn, uint32_t treelet, std::vector<uint32_t>& interior_index, uint32_t& n_interior, unsigned threads = 0)
{
    const uint32_t NONE = 0xFFFFFFFFu;
    interior_index.assign(nn, NONE);
    n_interior = 0;
    if (nn == 0) return 0;
    auto is_interior = [&](uint32_t i) { return (nodes[i].num_primitives_axis >> 16) == 0; };
    const uint32_t kTreelet = treelet ? treelet : 1u;
    // (a pool from 8 M nodes up: below, the one-thread walk is 0.03 s and the upload's other threads -- the own tree's builders start at the same moment -- are
    // better served by the cores: 0.050 instead of 0.030 s on the 4.9 M-node headline scene, 0.047 instead of 0.119 s on the 17.5 M-node one, profiles/r06/call50*.log)
    const unsigned T = threads ? threads : (nn >= 8000000u ? std::max(1u, std::min(std::thread::hardware_concurrency(), 16u)) : 1u);
    struct Met { uint32_t what; bool sub; };
    // the walk from `root`: interior nodes in record order to `list`; with cut != 0, a cluster root at cluster depth `cut` is noted as a subtree instead of being walked
    auto walk = [&](uint32_t root, uint32_t cut, std::vector<uint32_t>* list, std::vector<Met>* met, std::vector<uint32_t>* subs, size_t limit) -> int
    {
        std::vector<uint32_t> roots{root}, depth{0u}, cluster, frontier;
        size_t mine = 0;
        while (!roots.empty())
        {
            const uint32_t r = roots.back(), d = depth.back();
            roots.pop_back(); depth.pop_back();
            if (cut != 0u && d == cut) { met->push_back(Met{(uint32_t)subs->size(), true}); subs->push_back(r); continue; }
            cluster.clear(); frontier.clear();
            cluster.push_back(r);
            for (size_t head = 0; head < cluster.size(); ++head)          // BFS inside the treelet
            {
                const uint32_t n = cluster[head];
                const uint32_t kids[2] = {n + 1u, nodes[n].offset};
                for (uint32_t c : kids)
                {
                    if (c >= nn || c <= n) return 1;
                    if (!is_interior(c)) continue;
                    if (cluster.size() < kTreelet) cluster.push_back(c); else frontier.push_back(c);
                }
            }
            mine += cluster.size();
            if (mine > limit) return 2;
            for (uint32_t n : cluster) { if (list) list->push_back(n); else met->push_back(Met{n, false}); }
            for (size_t k = frontier.size(); k-- > 0;) { roots.push_back(frontier[k]); depth.push_back(d + 1u); }   // first child's cluster next
        }
        return 0;
    };
    if (!is_interior(0)) return 0;
    if (T <= 1u)
    {
        std::vector<uint32_t> list;
        const int rc = walk(0u, 0u, &list, nullptr, nullptr, nn);
        if (rc) return rc;
        for (uint32_t n : list) { if (interior_index[n] != NONE) return 2; interior_index[n] = n_interior++; }
        return 0;
    }
    std::vector<Met> met;
    std::vector<uint32_t> subs;
    int rc = walk(0u, 4u, nullptr, &met, &subs, nn);
    if (rc) return rc;
    std::vector<std::vector<uint32_t>> lists(subs.size());
    std::atomic<int> bad{0};
    std::atomic<size_t> next{0};
    auto run = [&]() { for (size_t k; (k = next.fetch_add(1)) < subs.size() && !bad.load(std::memory_order_relaxed);) { const int e = walk(subs[k], 0u, &lists[k], nullptr, nullptr, nn); if (e) bad.store(e); } };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T && t < subs.size(); ++t) pool.emplace_back(run);
    run();
    for (auto& th : pool) th.join();
    if (bad.load()) return bad.load();
    size_t total = 0;
    for (const Met& m : met) total += m.sub ? lists[m.what].size() : 1u;
    if (total > nn) return 2;
    std::vector<size_t> first(subs.size(), 0);
    size_t at = 0;
    for (const Met& m : met)
    {
        if (!m.sub) { if (interior_index[m.what] != NONE) return 2; interior_index[m.what] = (uint32_t)at++; continue; }
        first[m.what] = at;
        at += lists[m.what].size();
    }
    n_interior = (uint32_t)at;
    // (a node shared by two subtrees would be numbered twice: noticed by the count above only if it inflates the total beyond nn -- the serial walk's own guard --
    // and otherwise by the exchange below)
    std::atomic<bool> twice{false};
    next.store(0);
    auto number = [&]() { for (size_t k; (k = next.fetch_add(1)) < subs.size();) for (size_t j = 0; j < lists[k].size(); ++j)
        if (__atomic_exchange_n(&interior_index[lists[k][j]], (uint32_t)(first[k] + j), __ATOMIC_RELAXED) != NONE) twice.store(true); };
    pool.clear();
    for (unsigned t = 1; t < T && t < subs.size(); ++t) pool.emplace_back(number);
    number();
    for (auto& th : pool) th.join();
    return twice.load() ? 2 : 0;
}
} // namespace treelet

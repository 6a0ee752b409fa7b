// build_wide_bvh's host passes on a pool (round 6: validation in slices, the dynamic programme over the index ranges below the top of the tree, the record
// numbering by subtrees): the records, their order and their roots must be those of one thread -- with measured weights, the surface area, a metric and the
// two-level collapse; on an array that is a tree but NOT in depth-first layout the range sweep must notice and fall back.  Under ThreadSanitizer (tests/test_own_tree_threads.py).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "rt_types.h"
#include "own_bvh.h"
#include "wide_bvh.h"
static std::mt19937 rng(77);
static float uf(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }
static std::vector<rt_bvh_node> tree_over(uint32_t n_leaves, bool breadth_first)
{
    std::vector<rt_bvh_node> nodes((size_t)2 * n_leaves - 1);
    struct Item { uint32_t b, e, pos; };
    std::vector<float> box((size_t)n_leaves * 6);
    for (uint32_t i = 0; i < n_leaves; ++i) { for (int a = 0; a < 3; ++a) { const float c = uf(-20, 20), h = uf(0.01f, 0.5f); box[6 * (size_t)i + a] = c - h; box[6 * (size_t)i + 3 + a] = c + h; } }
    if (!breadth_first)
    {
        // the reference's layout through own_bvh.h (depth first); its leaves are copies of these
        std::vector<Item> todo{{0, n_leaves, 0}};
        while (!todo.empty())
        {
            const Item it = todo.back(); todo.pop_back();
            rt_bvh_node n; memset(&n, 0, sizeof(n));
            if (it.e - it.b == 1)
            {
                const float* b = &box[6 * (size_t)it.b];
                n.bounds_min.x = b[0]; n.bounds_min.y = b[1]; n.bounds_min.z = b[2]; n.bounds_max.x = b[3]; n.bounds_max.y = b[4]; n.bounds_max.z = b[5];
                n.offset = it.b * 3u; n.num_primitives_axis = (3u << 16); nodes[it.pos] = n; continue;
            }
            const uint32_t mid = it.b + (it.e - it.b) / 2u, nl = mid - it.b;
            n.offset = it.pos + 2u * nl; nodes[it.pos] = n;
            todo.push_back({it.b, mid, it.pos + 1u}); todo.push_back({mid, it.e, it.pos + 2u * nl});
        }
        ownbvh::Metric m; std::vector<rt_bvh_node> out;
        if (!ownbvh::build(nodes.data(), (uint32_t)nodes.size(), m, out)) out.clear();
        return out;
    }
    return nodes;
}
static bool same(const std::vector<WideNode>& a, const std::vector<WideNode>& b, const std::vector<uint32_t>& ra, const std::vector<uint32_t>& rb)
{
    return a.size() == b.size() && ra == rb && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(WideNode)) == 0);
}
int main(int argc, char** argv)
{
    const uint32_t n_leaves = argc > 1 ? (uint32_t)atoi(argv[1]) : 160000u;
    const std::vector<rt_bvh_node> tree = tree_over(n_leaves, false);
    if (tree.empty()) { printf("FAIL: no tree\n"); return 1; }
    std::vector<double> w(tree.size()); for (auto& x : w) x = uf(0, 100);
    ownbvh::Metric mm; mm.iso = 0.5; mm.dirs.push_back({0.2, 0.4, 0.89});
    int checked = 0;
    for (int mode = 0; mode < 4; ++mode)
    {
        std::vector<WideNode> one, many; std::vector<uint32_t> r1, rn; uint32_t e1 = 0, en = 0;
        auto run = [&](unsigned threads, std::vector<WideNode>& out, uint32_t& entry, std::vector<uint32_t>& roots)
        {
            return mode == 0 ? rtw::build_wide_bvh(tree.data(), (uint32_t)tree.size(), rtw::RT_WIDE_SAH, out, entry, &roots, nullptr, w.data(), nullptr, threads)
                 : mode == 1 ? rtw::build_wide_bvh(tree.data(), (uint32_t)tree.size(), rtw::RT_WIDE_SAH, out, entry, &roots, nullptr, nullptr, nullptr, threads)
                 : mode == 2 ? rtw::build_wide_bvh(tree.data(), (uint32_t)tree.size(), rtw::RT_WIDE_SAH, out, entry, &roots, &mm, nullptr, nullptr, threads)
                             : rtw::build_wide_bvh(tree.data(), (uint32_t)tree.size(), rtw::RT_WIDE_TWO_LEVELS, out, entry, &roots, nullptr, nullptr, nullptr, threads);
        };
        if (!run(1, one, e1, r1) || one.empty()) { printf("FAIL: one thread, mode %d\n", mode); return 1; }
        for (unsigned threads : {2u, 5u, 16u})
        {
            if (!run(threads, many, en, rn)) { printf("FAIL: %u threads, mode %d\n", threads, mode); return 1; }
            if (e1 != en || !same(one, many, r1, rn)) { printf("FAIL: %u threads fold differently (mode %d)\n", threads, mode); return 1; }
            ++checked;
        }
    }
    {
        // A tree that is NOT in depth-first layout: root 0 = (1, 3), node 1 = (leaf 2, the LAST node of the array), the big tree above shifted to [3, 3 + S) in between.
        // Children follow their parents and the boxes nest, so it is a valid input; node 1's second child lies outside the range [1, 3) the sweep gives it:
        // the sweep must notice and the fold must be the one-thread fold.
        const uint32_t S = (uint32_t)tree.size();
        std::vector<rt_bvh_node> odd((size_t)S + 4u);
        for (uint32_t i = 0; i < S; ++i) { odd[3u + i] = tree[i]; if ((tree[i].num_primitives_axis >> 16) == 0) odd[3u + i].offset += 3u; }
        auto leaf = [&](float c, uint32_t first) { rt_bvh_node n; memset(&n, 0, sizeof(n)); n.bounds_min.x = n.bounds_min.y = n.bounds_min.z = c; n.bounds_max.x = n.bounds_max.y = n.bounds_max.z = c + 0.5f;
                                                    n.offset = first; n.num_primitives_axis = (3u << 16); return n; };
        odd[2] = leaf(30.0f, 3u * n_leaves); odd[S + 3u] = leaf(-31.0f, 3u * n_leaves + 3u);
        auto join = [&](const rt_bvh_node& a, const rt_bvh_node& b, uint32_t second) { rt_bvh_node n; memset(&n, 0, sizeof(n));
            n.bounds_min.x = std::min(a.bounds_min.x, b.bounds_min.x); n.bounds_min.y = std::min(a.bounds_min.y, b.bounds_min.y); n.bounds_min.z = std::min(a.bounds_min.z, b.bounds_min.z);
            n.bounds_max.x = std::max(a.bounds_max.x, b.bounds_max.x); n.bounds_max.y = std::max(a.bounds_max.y, b.bounds_max.y); n.bounds_max.z = std::max(a.bounds_max.z, b.bounds_max.z);
            n.offset = second; return n; };
        odd[1] = join(odd[2], odd[S + 3u], S + 3u);
        odd[0] = join(odd[1], odd[3], 3u);
        std::vector<WideNode> one, many; std::vector<uint32_t> r1, rn; uint32_t e1 = 0, en = 0;
        const bool ok1 = rtw::build_wide_bvh(odd.data(), (uint32_t)odd.size(), rtw::RT_WIDE_SAH, one, e1, &r1, nullptr, nullptr, nullptr, 1);
        const bool okn = rtw::build_wide_bvh(odd.data(), (uint32_t)odd.size(), rtw::RT_WIDE_SAH, many, en, &rn, nullptr, nullptr, nullptr, 4);
        if (!ok1 || !okn || e1 != en || !same(one, many, r1, rn)) { printf("FAIL: the out-of-layout tree folds differently on 4 threads (%d / %d)\n", (int)ok1, (int)okn); return 1; }
        ++checked;
    }
    printf("ok: %d folds identical to the one-thread fold\n", checked);
    return 0;
}

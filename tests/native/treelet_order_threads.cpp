// treelet_order.h on a pool: the record order must be the one-thread order for every thread count and treelet size, and arrays that are not trees must be refused
// the same way (tests/test_own_tree_threads.py: ThreadSanitizer and AddressSanitizer + UBSan).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "rt_types.h"
#include "own_bvh.h"
#include "treelet_order.h"
static std::mt19937 rng(31);
static float uf(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }
int main(int argc, char** argv)
{
    const uint32_t n_leaves = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;
    std::vector<rt_bvh_node> flat((size_t)2 * n_leaves - 1);
    struct Item { uint32_t b, e, pos; };
    std::vector<Item> todo{{0, n_leaves, 0}};
    while (!todo.empty())
    {
        const Item it = todo.back(); todo.pop_back();
        rt_bvh_node n; memset(&n, 0, sizeof(n));
        if (it.e - it.b == 1)
        {
            const float c[3] = {uf(-20, 20), uf(-20, 20), uf(0, 6)}, h = uf(0.01f, 0.4f);
            n.bounds_min.x = c[0] - h; n.bounds_min.y = c[1] - h; n.bounds_min.z = c[2] - h; n.bounds_max.x = c[0] + h; n.bounds_max.y = c[1] + h; n.bounds_max.z = c[2] + h;
            n.offset = it.b * 3u; n.num_primitives_axis = (3u << 16); flat[it.pos] = n; continue;
        }
        const uint32_t mid = it.b + (it.e - it.b) / 2u, nl = mid - it.b;
        n.offset = it.pos + 2u * nl; flat[it.pos] = n;
        todo.push_back({it.b, mid, it.pos + 1u}); todo.push_back({mid, it.e, it.pos + 2u * nl});
    }
    ownbvh::Metric m; std::vector<rt_bvh_node> tree;                      // an SAH tree: lopsided where the leaves cluster
    if (!ownbvh::build(flat.data(), (uint32_t)flat.size(), m, tree)) { printf("FAIL: no tree\n"); return 1; }
    int checked = 0;
    for (uint32_t treelet : {1u, 3u, 7u, 15u})
    {
        std::vector<uint32_t> one, many; uint32_t n1 = 0, nm = 0;
        if (treelet::order(tree.data(), (uint32_t)tree.size(), treelet, one, n1, 1) != 0 || n1 != n_leaves - 1u) { printf("FAIL: one thread (treelet %u)\n", treelet); return 1; }
        std::vector<uint8_t> seen(n1, 0);
        for (uint32_t i = 0; i < tree.size(); ++i) if ((tree[i].num_primitives_axis >> 16) == 0) { if (one[i] >= n1 || seen[one[i]]) { printf("FAIL: not a permutation\n"); return 1; } seen[one[i]] = 1; }
        for (unsigned threads : {2u, 5u, 16u})
        {
            if (treelet::order(tree.data(), (uint32_t)tree.size(), treelet, many, nm, threads) != 0 || nm != n1 || many != one) { printf("FAIL: %u threads order differently (treelet %u)\n", threads, treelet); return 1; }
            ++checked;
        }
    }
    {
        // a shared child and a child index behind the array: refused on every thread count
        std::vector<rt_bvh_node> dag = tree; std::vector<uint32_t> idx; uint32_t n = 0;
        uint32_t victim = 0; for (uint32_t i = 1; i < dag.size(); ++i) if ((dag[i].num_primitives_axis >> 16) == 0 && (dag[dag[i].offset].num_primitives_axis >> 16) == 0 && (dag[i + 1].num_primitives_axis >> 16) == 0 && dag[i].offset - i > 64u) { victim = i; break; }
        // `other`: an interior node of victim's FIRST subtree (it stays reachable) whose second child is interior -- that child gets victim as a second parent
        uint32_t other = 0; for (uint32_t i = victim + 1; i < dag[victim].offset; ++i) if ((dag[i].num_primitives_axis >> 16) == 0 && (dag[dag[i].offset].num_primitives_axis >> 16) == 0) { other = i; break; }
        if (victim && other) { dag[victim].offset = dag[other].offset; for (unsigned threads : {1u, 4u}) if (treelet::order(dag.data(), (uint32_t)dag.size(), 7u, idx, n, threads) == 0) { printf("FAIL: a shared child passed on %u threads\n", threads); return 1; } ++checked; }
        std::vector<rt_bvh_node> out_of = tree; out_of[victim].offset = (uint32_t)tree.size() + 5u;
        for (unsigned threads : {1u, 4u}) if (treelet::order(out_of.data(), (uint32_t)out_of.size(), 7u, idx, n, threads) != 1) { printf("FAIL: a child index behind the array passed on %u threads\n", threads); return 1; }
        ++checked;
    }
    printf("ok: %d orders identical to the one-thread order\n", checked);
    return 0;
}

// own_bvh.h's parallel build: the tree must not depend on the number of threads, a raised cancel flag must end the build, and none of it may race
// (compiled twice by tests/test_own_tree_threads.py: ThreadSanitizer, and AddressSanitizer + UBSan).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "rt_types.h"
#include "own_bvh.h"
static std::mt19937 rng(4242);
static float uf(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }
struct Box { float mn[3], mx[3]; };
// a balanced "reference" tree in the linear layout over n random leaf boxes (clustered, some coincident centres)
static std::vector<rt_bvh_node> reference_tree(uint32_t n_leaves, int flavour)
{
    std::vector<Box> leaves(n_leaves);
    for (uint32_t i = 0; i < n_leaves; ++i)
    {
        Box& b = leaves[i];
        const float cluster[3] = {uf(-50, 50), uf(-50, 50), uf(0, 10)};
        for (int a = 0; a < 3; ++a)
        {
            float c = flavour == 2 ? (float)(i % 7) : cluster[a] + uf(-1, 1), h = flavour == 1 ? 0.0f : uf(0, 0.5f);
            if (flavour == 2 && a == 2) c = 1.0f;                                        // many coincident centroids: the "halve by reference order" branch
            b.mn[a] = c - h; b.mx[a] = c + h;
        }
    }
    std::vector<rt_bvh_node> nodes((size_t)2 * n_leaves - 1);
    struct Item { uint32_t b, e, pos; };
    std::vector<Item> todo{{0, n_leaves, 0}};
    while (!todo.empty())
    {
        const Item it = todo.back(); todo.pop_back();
        rt_bvh_node n; memset(&n, 0, sizeof(n));
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t i = it.b; i < it.e; ++i) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], leaves[i].mn[a]); mx[a] = std::max(mx[a], leaves[i].mx[a]); }
        n.bounds_min.x = mn[0]; n.bounds_min.y = mn[1]; n.bounds_min.z = mn[2]; n.bounds_max.x = mx[0]; n.bounds_max.y = mx[1]; n.bounds_max.z = mx[2];
        if (it.e - it.b == 1) { n.offset = it.b * 3u; n.num_primitives_axis = (3u << 16); nodes[it.pos] = n; continue; }
        const uint32_t mid = it.b + (it.e - it.b) / 2u, nl = mid - it.b;
        n.offset = it.pos + 2u * nl; n.num_primitives_axis = it.pos % 3u;
        nodes[it.pos] = n;
        todo.push_back({it.b, mid, it.pos + 1u});
        todo.push_back({mid, it.e, it.pos + 2u * nl});
    }
    return nodes;
}
int main(int argc, char** argv)
{
    const uint32_t big = argc > 1 ? (uint32_t)atoi(argv[1]) : 150000u;
    ownbvh::Metric m; m.iso = 0.5; m.dirs.push_back({0.2, 0.4, 0.89});
    int trees = 0;
    for (int flavour = 0; flavour < 3; ++flavour)
        for (uint32_t n_leaves : {2u, 3u, 1000u, 40000u, big})
        {
            if (flavour == 2 && n_leaves > 40000u) continue;
            const std::vector<rt_bvh_node> ref = reference_tree(n_leaves, flavour);
            std::vector<rt_bvh_node> one, many;
            if (!ownbvh::build(ref.data(), (uint32_t)ref.size(), m, one, nullptr, 1)) { printf("FAIL: build with one thread, %u leaves\n", n_leaves); return 1; }
            if (one.size() != ref.size()) { printf("FAIL: size\n"); return 1; }
            for (unsigned threads : {2u, 3u, 8u, 32u})
            {
                if (!ownbvh::build(ref.data(), (uint32_t)ref.size(), m, many, nullptr, threads)) { printf("FAIL: build with %u threads\n", threads); return 1; }
                if (many.size() != one.size() || memcmp(many.data(), one.data(), one.size() * sizeof(rt_bvh_node)) != 0)
                { printf("FAIL: %u threads build another tree than one thread (%u leaves, flavour %d)\n", threads, n_leaves, flavour); return 1; }
            }
            ++trees;
        }
    // a cancel raised while the pool works: build() returns false and does not hang; raised before: the same
    {
        const std::vector<rt_bvh_node> ref = reference_tree(big, 0);
        for (int delay_us : {0, 200, 2000, 20000})
        {
            std::atomic<bool> cancel{delay_us == 0};
            std::thread raiser([&] { std::this_thread::sleep_for(std::chrono::microseconds(delay_us)); cancel.store(true); });
            std::vector<rt_bvh_node> out;
            const bool ok = ownbvh::build(ref.data(), (uint32_t)ref.size(), m, out, &cancel, 8);
            raiser.join();
            if (ok && out.size() != ref.size()) { printf("FAIL: a finished build of the wrong size\n"); return 1; }
            if (!ok && !out.empty()) { printf("FAIL: a cancelled build left a tree behind\n"); return 1; }
            if (delay_us == 0 && ok) { printf("FAIL: a build cancelled before it began returned a tree\n"); return 1; }
        }
    }
    printf("ok: %d trees identical for 1, 2, 3, 8 and 32 threads; cancelled builds end\n", trees);
    return 0;
}

// What own_bvh.h's build costs on this host by thread count and phase, over N clustered random leaves (no GPU; the trees are compared byte for byte on the way).
// Build: g++ -std=c++17 -O2 -pthread -I include -I raytracing_amd/csrc tools/own_bvh_bench.cpp -o tools/bin/own_bvh_bench        Run: tools/bin/own_bvh_bench [leaves = 8700000]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "rt_types.h"
#include "own_bvh.h"
static std::mt19937 rng(99);
static float uf(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const uint32_t n_leaves = argc > 1 ? (uint32_t)atoi(argv[1]) : 8700000u;
    std::vector<rt_bvh_node> nodes((size_t)2 * n_leaves - 1);
    struct Item { uint32_t b, e, pos; };
    std::vector<Item> todo{{0, n_leaves, 0}};
    while (!todo.empty())           // a balanced tree in the reference's linear layout; its interior boxes are never read by the builder
    {
        const Item it = todo.back(); todo.pop_back();
        rt_bvh_node n; memset(&n, 0, sizeof(n));
        if (it.e - it.b == 1)
        {
            const float cl[3] = {uf(-30, 30), uf(-30, 30), uf(0, 8)};
            float c[3], h[3]; for (int a = 0; a < 3; ++a) { c[a] = cl[a] + uf(-1, 1); h[a] = uf(0, 0.4f); }
            n.bounds_min.x = c[0] - h[0]; n.bounds_min.y = c[1] - h[1]; n.bounds_min.z = c[2] - h[2]; n.bounds_max.x = c[0] + h[0]; n.bounds_max.y = c[1] + h[1]; n.bounds_max.z = c[2] + h[2];
            n.offset = it.b * 3u; n.num_primitives_axis = (3u << 16); nodes[it.pos] = n; continue;
        }
        const uint32_t mid = it.b + (it.e - it.b) / 2u, nl = mid - it.b;
        n.offset = it.pos + 2u * nl; n.num_primitives_axis = it.pos % 3u; nodes[it.pos] = n;
        todo.push_back({it.b, mid, it.pos + 1u}); todo.push_back({mid, it.e, it.pos + 2u * nl});
    }
    ownbvh::Metric m; m.iso = 0.5; m.dirs.push_back({0.2, 0.4, 0.89});
    printf("%u leaves, %u hardware threads\n", n_leaves, std::thread::hardware_concurrency());
    std::vector<rt_bvh_node> first;
    for (unsigned th : {32u, 16u, 8u, 4u, 1u})
    {
        std::vector<rt_bvh_node> out; double ph[3] = {0, 0, 0}; const double t0 = now();
        const bool ok = ownbvh::build(nodes.data(), (uint32_t)nodes.size(), m, out, nullptr, th, ph);
        const double t = now() - t0;
        if (first.empty()) first = out;
        printf("%2u threads: %.3f s = collecting the leaves %.3f + allocating %.3f + the pool %.3f (%s, %s)\n", th, t, ph[0], ph[1], ph[2], ok ? "ok" : "FAILED",
            out.size() == first.size() && memcmp(out.data(), first.data(), out.size() * sizeof(rt_bvh_node)) == 0 ? "the same tree" : "ANOTHER TREE");
    }
    return 0;
}

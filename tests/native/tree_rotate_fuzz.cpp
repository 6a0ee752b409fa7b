// ASAN / UBSAN harness for tree_rotate.h: random binary trees over random leaf boxes, random rays; invariants after rotation.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include <functional>
#include "rt_types.h"
#include "tree_rotate.h"
static std::mt19937 rng(12345);
static float uf(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }
struct Box { float mn[3], mx[3]; };
int main()
{
    for (int iter = 0; iter < 400; ++iter)
    {
        const int n_leaves = 1 + (int)(rng() % 200);
        std::vector<Box> leaves(n_leaves);
        for (auto& b : leaves) for (int a = 0; a < 3; ++a) { float c = uf(-10, 10), h = uf(0, iter % 3 ? 1.5f : 0.0f); b.mn[a] = c - h; b.mx[a] = c + h; }
        std::vector<rt_bvh_node> nodes((size_t)2 * n_leaves - 1);
        std::vector<int> idx(n_leaves); for (int i = 0; i < n_leaves; ++i) idx[i] = i;
        std::function<uint32_t(int, int, uint32_t)> build = [&](int b, int e, uint32_t pos) -> uint32_t
        {
            rt_bvh_node n; memset(&n, 0, sizeof(n));
            float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
            for (int i = b; i < e; ++i) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], leaves[idx[i]].mn[a]); mx[a] = std::max(mx[a], leaves[idx[i]].mx[a]); }
            n.bounds_min.x = mn[0]; n.bounds_min.y = mn[1]; n.bounds_min.z = mn[2]; n.bounds_max.x = mx[0]; n.bounds_max.y = mx[1]; n.bounds_max.z = mx[2];
            if (e - b == 1) { n.offset = (uint32_t)idx[b] * 3u; n.num_primitives_axis = (3u << 16); nodes[pos] = n; return 1; }
            const int mid = b + 1 + (int)(rng() % (e - b - 1));            // lopsided on purpose
            const uint32_t sl = build(b, mid, pos + 1);
            n.offset = pos + 1 + sl; n.num_primitives_axis = rng() % 3;
            const uint32_t sr = build(mid, e, pos + 1 + sl);
            nodes[pos] = n;
            return 1 + sl + sr;
        };
        std::shuffle(idx.begin(), idx.end(), rng);
        build(0, n_leaves, 0);
        const size_t n_rays = rng() % 300;
        std::vector<float> o(4 * n_rays + 4), d(4 * n_rays + 4);
        for (size_t r = 0; r < n_rays; ++r)
        {
            for (int a = 0; a < 3; ++a) { o[4 * r + a] = uf(-12, 12); d[4 * r + a] = (rng() % 11 == 0) ? 0.0f : uf(-1, 1); }
            o[4 * r + 3] = (rng() % 7 == 0) ? 0.0f : uf(0, 50);
            if (rng() % 53 == 0) o[4 * r] = NAN;
        }
        std::vector<rt_bvh_node> out;
        double cost[2];
        const int moves = 1 + (int)(rng() % 3);
        const int passes = 1 + (int)(rng() % 8);
        const double min_gain = (iter % 2) ? 0.0 : 0.03;
        const uint32_t made = treerot::rotate(nodes.data(), (uint32_t)nodes.size(), o.data(), d.data(), n_rays, passes, out, cost, nullptr, moves, min_gain, nullptr, 1);
        {
            // the passes on several threads (subtrees with a list of `grain` rays or more go to the shared queue): the same rotations, the same tree
            std::vector<rt_bvh_node> out_mt;
            double cost_mt[2];
            const unsigned threads = 2 + (unsigned)(rng() % 7);
            const size_t grain = 1 + (size_t)(rng() % 40);
            const uint32_t made_mt = treerot::rotate(nodes.data(), (uint32_t)nodes.size(), o.data(), d.data(), n_rays, passes, out_mt, cost_mt, nullptr, moves, min_gain, nullptr, threads, grain);
            if (made_mt != made || out_mt.size() != out.size() || (!out.empty() && memcmp(out_mt.data(), out.data(), out.size() * sizeof(rt_bvh_node)) != 0) || cost_mt[1] != cost[1])
            { printf("FAIL: %u threads (grain %zu) rotate differently: %u / %u rotations, iter %d\n", threads, grain, made_mt, made, iter); return 1; }
        }
        if (n_rays == 0) { if (!out.empty() || made) { printf("FAIL: no rays\n"); return 1; } continue; }
        if (out.size() != nodes.size()) { printf("FAIL size %zu %zu iter %d\n", out.size(), nodes.size(), iter); return 1; }
        if (cost[1] > cost[0] + 1e-9) { printf("FAIL cost went up\n"); return 1; }
        // same leaves, exact unions, layout
        std::vector<uint32_t> seen;
        std::function<uint32_t(uint32_t)> check = [&](uint32_t i) -> uint32_t
        {
            const rt_bvh_node& n = out[i];
            if ((n.num_primitives_axis >> 16) != 0) { seen.push_back(n.offset); return 1; }
            if ((n.num_primitives_axis & 0xFFFF) > 2) { printf("FAIL axis\n"); exit(1); }
            const uint32_t a = i + 1, b = n.offset;
            if (b <= a || b >= out.size()) { printf("FAIL child index\n"); exit(1); }
            const uint32_t sa = check(a);
            if (b != a + sa) { printf("FAIL layout\n"); exit(1); }
            const uint32_t sb = check(b);
            const float* amn = &out[a].bounds_min.x; const float* amx = &out[a].bounds_max.x; const float* bmn = &out[b].bounds_min.x; const float* bmx = &out[b].bounds_max.x;
            const float* nmn = &n.bounds_min.x; const float* nmx = &n.bounds_max.x;
            for (int k = 0; k < 3; ++k) if (nmn[k] != std::min(amn[k], bmn[k]) || nmx[k] != std::max(amx[k], bmx[k])) { printf("FAIL union\n"); exit(1); }
            return 1 + sa + sb;
        };
        if (check(0) != out.size()) { printf("FAIL count\n"); return 1; }
        std::sort(seen.begin(), seen.end());
        for (int i = 0; i < n_leaves; ++i) if (seen[i] != (uint32_t)i * 3u) { printf("FAIL leaves\n"); return 1; }
    }
    printf("ok: 400 random trees\n");
    return 0;
}

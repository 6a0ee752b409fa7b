// tree_select.h -- which 4-wide tree should a ray population walk?  Measured, not assumed (host code; rt_scene_upload, round 4).
//
// own_bvh.h can build several binary trees over the reference's leaves (surface-area metric, projected area along the
// directional lights with more or less isotropy) and the reference's own topology is a candidate too.  None wins everywhere:
// on the 2.8 M-triangle city block the own trees save 5 % of a shadow ray's steps, on the Cornell shell + 0.9 M-triangle blob
// the projected-area tree saves 33 %, in dense foliage the reference's topology is as good as any (tools/own_tree_study.py).
// So rt_scene_upload builds the candidates and WALKS each with a few thousand proxy rays of the population it is for:
//   shadow   origins spread over the scene's surfaces by area (+ the reference's 1e-3 offset along the normal, towards the
//            light), one ray per analytic light in turn, any-hit;
//   closest  the same origins, cosine-distributed directions about the normal: the bounce rays of a path tracer.
// The walk is k_trace_w4's (wide-node visits + leaf arrivals + further triangles = its loop passes) in plain float on the
// dequantised boxes -- a COST estimate: nothing here decides a hit, the kernel does that exactly on whichever tree wins.
#pragma once
#include <stdint.h>
#include <math.h>
#include <vector>
#include "rt_types.h"

namespace treesel
{
struct ProxyRay { float o[3], d[3], t_max; };

struct Rng
{
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    double uni() { return (next() + 0.5) / 2147483648.0; }
};

// `count` surface points by area; rays towards the lights (shadow) or cosine-distributed about the normal (closest)
inline std::vector<ProxyRay> proxy_rays(const rt_triangle* tris, uint32_t nt, const rt_light* lights, uint32_t n_lights, uint32_t count,
    bool shadow, uint64_t seed = 4)
{
    std::vector<ProxyRay> rays;
    if (nt == 0 || (shadow && n_lights == 0)) return rays;
    std::vector<double> cdf(nt);
    double total = 0.0;
    auto P = [&](const rt_vertex& v) { return v.position; };
    for (uint32_t i = 0; i < nt; ++i)
    {
        const rt_float3 a = P(tris[i].v1), b = P(tris[i].v2), c = P(tris[i].v3);
        const double e1[3] = {(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z}, e2[3] = {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double area = 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        total += std::isfinite(area) ? area : 0.0;
        cdf[i] = total;
    }
    if (!(total > 0.0)) return rays;
    Rng rng(seed);
    rays.reserve(count);
    for (uint32_t k = 0; k < count; ++k)
    {
        const double x = rng.uni() * total;
        uint32_t lo = 0, hi = nt - 1;
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (cdf[mid] < x) lo = mid + 1; else hi = mid; }
        const rt_float3 a = P(tris[lo].v1), b = P(tris[lo].v2), c = P(tris[lo].v3);
        double u = rng.uni(), v = rng.uni();
        if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
        const double p[3] = {a.x + u * ((double)b.x - a.x) + v * ((double)c.x - a.x), a.y + u * ((double)b.y - a.y) + v * ((double)c.y - a.y),
                             a.z + u * ((double)b.z - a.z) + v * ((double)c.z - a.z)};
        const double e1[3] = {(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z}, e2[3] = {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nl > 0.0)) continue;
        for (double& q : n) q /= nl;
        double d[3];
        double t_max = 20000.0;                                              // constants.h:28
        if (shadow)
        {
            const rt_light& l = lights[k % n_lights];
            if (l.type == RT_LIGHT_TYPE_POINT) { d[0] = l.origin.x - p[0]; d[1] = l.origin.y - p[1]; d[2] = l.origin.z - p[2]; }
            else { d[0] = l.origin.x; d[1] = l.origin.y; d[2] = l.origin.z; }
            const double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (!(dl > 0.0)) continue;
            for (double& q : d) q /= dl;
            t_max = l.type == RT_LIGHT_TYPE_POINT ? dl : 20000.0;
            if (n[0] * d[0] + n[1] * d[1] + n[2] * d[2] < 0.0) for (double& q : n) q = -q;   // the lit side
        }
        else
        {
            if (rng.next() & 1u) for (double& q : n) q = -q;
            // cosine-distributed direction about n
            const double r1 = rng.uni(), r2 = rng.uni(), r = std::sqrt(r1), phi = 6.283185307179586 * r2;
            double t1[3] = {std::fabs(n[0]) < 0.9 ? 1.0 : 0.0, std::fabs(n[0]) < 0.9 ? 0.0 : 1.0, 0.0};
            double b1[3] = {t1[1] * n[2] - t1[2] * n[1], t1[2] * n[0] - t1[0] * n[2], t1[0] * n[1] - t1[1] * n[0]};
            const double bl = std::sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
            for (double& q : b1) q /= bl;
            const double b2[3] = {n[1] * b1[2] - n[2] * b1[1], n[2] * b1[0] - n[0] * b1[2], n[0] * b1[1] - n[1] * b1[0]};
            const double z = std::sqrt(std::max(0.0, 1.0 - r1));
            for (int q = 0; q < 3; ++q) d[q] = r * std::cos(phi) * b1[q] + r * std::sin(phi) * b2[q] + z * n[q];
        }
        ProxyRay ray;
        for (int q = 0; q < 3; ++q) { ray.o[q] = (float)(p[q] + 1e-3 * n[q]); ray.d[q] = (float)d[q]; }
        ray.t_max = (float)t_max;
        rays.push_back(ray);
    }
    return rays;
}

// 64-byte wide record as build_wide_bvh writes it (rt_hip.hip: WideNode)
struct Record { float ox, oy, oz; uint32_t meta; uint32_t lo[3]; uint32_t hi[3]; uint32_t ref[4]; uint32_t order; uint32_t pad; };

// mean steps per ray (wide-node visits + leaf arrivals + further triangles) of the proxy rays over one tree
inline double walk_cost(const Record* rec, uint32_t n_rec, uint32_t entry, const rt_bvh_node* ref_nodes, const uint32_t* leaf_of_first /* [nt]: leaf node of a first triangle */,
    const rt_triangle* tris, const std::vector<ProxyRay>& rays, bool any_hit)
{
    if (rays.empty()) return 0.0;
    uint64_t steps = 0;
    std::vector<std::pair<uint32_t, float>> stack;
    for (const ProxyRay& r : rays)
    {
        const float inv[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
        float t_max = r.t_max;
        auto slab = [&](const float mn[3], const float mx[3], float& entry_t)
        {
            float t0 = 0.0f, t1 = t_max;
            for (int a = 0; a < 3; ++a)
            {
                float ta = (mn[a] - r.o[a]) * inv[a], tb = (mx[a] - r.o[a]) * inv[a];
                if (ta > tb) { const float s = ta; ta = tb; tb = s; }
                if (ta > t0) t0 = ta;                                        // (NaNs compare false: the plane is ignored -- an estimate)
                if (tb < t1) t1 = tb;
            }
            entry_t = t0;
            return t1 >= t0;
        };
        stack.clear();
        uint32_t ref = entry;
        bool have = true, done = false;
        while (have && !done)
        {
            if (ref & 0x80000000u)
            {
                ++steps;
                const uint32_t first = ref & 0x7FFFFFFFu;
                const rt_bvh_node& L = ref_nodes[leaf_of_first[first]];
                const float mn[3] = {L.bounds_min.x, L.bounds_min.y, L.bounds_min.z}, mx[3] = {L.bounds_max.x, L.bounds_max.y, L.bounds_max.z};
                float e;
                if (slab(mn, mx, e))
                {
                    const uint32_t np = L.num_primitives_axis >> 16;
                    steps += np - 1u;
                    for (uint32_t i = 0; i < np && !done; ++i)
                    {
                        const rt_triangle& T = tris[first + i];
                        const float p1[3] = {T.v1.position.x, T.v1.position.y, T.v1.position.z};
                        const float e1[3] = {T.v2.position.x - p1[0], T.v2.position.y - p1[1], T.v2.position.z - p1[2]};
                        const float e2[3] = {T.v3.position.x - p1[0], T.v3.position.y - p1[1], T.v3.position.z - p1[2]};
                        const float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
                        const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                        if (!(det > 1e-8f)) continue;
                        const float id = 1.0f / det;
                        const float tv[3] = {r.o[0] - p1[0], r.o[1] - p1[1], r.o[2] - p1[2]};
                        const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * id;
                        if (u < 0.0f || u > 1.0f) continue;
                        const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                        const float v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * id;
                        if (v < 0.0f || u + v > 1.0f) continue;
                        const float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * id;
                        if (t < 0.0f || t > t_max) continue;
                        t_max = t;
                        if (any_hit) { done = true; steps -= np - 1u - i; }
                    }
                }
            }
            else if (ref < n_rec)
            {
                ++steps;
                const Record& n = rec[ref];
                const float cell[3] = {ldexpf(1.0f, (int)(n.meta & 0xFFu) - 127), ldexpf(1.0f, (int)((n.meta >> 8) & 0xFFu) - 127), ldexpf(1.0f, (int)((n.meta >> 16) & 0xFFu) - 127)};
                const float org[3] = {n.ox, n.oy, n.oz};
                uint32_t pr[4]; float pe[4]; int np = 0;
                for (int k = 0; k < 4; ++k)
                {
                    if (n.ref[k] == 0xFFFFFFFFu) continue;
                    float mn[3], mx[3];
                    for (int a = 0; a < 3; ++a)
                    {
                        mn[a] = org[a] + (float)((n.lo[a] >> (8 * k)) & 0xFFu) * cell[a];
                        mx[a] = org[a] + (float)((n.hi[a] >> (8 * k)) & 0xFFu) * cell[a];
                    }
                    float e;
                    if (slab(mn, mx, e)) { pr[np] = n.ref[k]; pe[np] = e; ++np; }
                }
                if (!any_hit)
                    for (int i = 1; i < np; ++i)                            // nearest first (a cost estimate: not the record's order table)
                        for (int j = i; j > 0 && pe[j] < pe[j - 1]; --j) { std::swap(pe[j], pe[j - 1]); std::swap(pr[j], pr[j - 1]); }
                for (int i = np - 1; i >= 1; --i) stack.push_back({pr[i], pe[i]});
                if (np) { ref = pr[0]; continue; }
            }
            have = false;
            while (!stack.empty())
            {
                const auto top = stack.back();
                stack.pop_back();
                if (any_hit || t_max >= top.second) { ref = top.first; have = true; break; }
            }
        }
    }
    return (double)steps / (double)rays.size();
}
} // namespace treesel

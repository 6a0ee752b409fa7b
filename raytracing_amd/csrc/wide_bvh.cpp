// wide_bvh.cpp -- the host side of the backend's tree work: see wide_bvh.h.  (Moved out of rt_hip.hip in round 6, unchanged.)
#include "wide_bvh.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <thread>
#include "rt_hip.h"

namespace rtw
{
// ---- 4-wide quantized BVH (k_trace_w4) --------------------------------------
// A connected piece of the reference BVH2 (LinearBVHNode[], bvh.cpp:223-245) -- a node and up to two more interior
// nodes below it -- is folded into one 64-byte record: up to four "slots" = the frontier of that piece (which
// frontier: `collapse`, below).  Slot boxes are stored as 8-bit grid coordinates relative to a per-node frame
// (origin, power-of-two cell size per axis), rounded OUTWARD.
//
// Why results stay bit-identical to the reference ("wide traversal"):
//  * the frame is chosen so that origin + q * cell is exactly representable in binary32 for every
//    q in 0..255 (origin is a multiple of the cell, |origin| / cell < 2^23), so the kernel
//    dequantises WITHOUT rounding and evaluates the reference's own expression
//    fl(fl(b - o) * inv) on a box that contains the true one; that expression is monotone in b,
//    hence "true box passes  =>  stored box passes": interior culling only ever visits MORE;
//  * every leaf is box-tested again with its exact fp32 bounds and the ray's current t_max when it
//    is reached (the bounds travel in the leaf's first triangle record), and leaves are reached in
//    the reference's depth-first near/far order (the slots are brought into that order per direction octant by
//    tabulated exchanges, see `arrange`).  Node bounds are exact unions of their children's
//    (bvh.hpp:73, checked below), so a leaf's box passing implies that all its ancestors' boxes
//    pass: the reference tests the triangles of a leaf iff that leaf's own box test passes at that
//    point of the traversal -- which is exactly what the kernel evaluates.
// Record: q0 = (origin.xyz, meta)   meta = ex | ey << 8 | ez << 16 | occupied slots << 24 (biased exponents of the cell sizes)
//         q1 = (lo.x, lo.y, lo.z, hi.x)    one byte per slot in every dword
//         q2 = (hi.y, hi.z, ref0, ref1)    ref = wide node index | RT_LEAF_BIT + first triangle | RT_EMPTY_REF
//         q3 = (ref2, ref3, order, -)       order: for each of the 8 direction-sign octants o (bit a set = direction negative
//                                          along axis a) four bits at 4 * o = the conditional exchanges of slots (0,1), (2,3),
//                                          (0,2), (1,3), made in that order, that bring the occupied slots into the
//                                          reference's visit order (see `arrange` in build_wide_bvh)
// (struct WideNode: fold_kernels.h -- the device builds the same records, device_fold.h)

// Which BVH2 nodes become the four slots of a record (`collapse`):
//  RT_WIDE_TWO_LEVELS  the grandchildren (a child that is a leaf fills one slot): round 2's rule;
//  RT_WIDE_SAH         the frontier that minimises the expected number of wide-node visits: a record rooted at BVH2 node n
//                      is visited when a ray passes n's slot box (probability ~ area(n)), the interior nodes between n and
//                      its slots are never tested at all, leaves are what they are -- so the cost of a collapse is the sum of
//                      area(root) over its records, minimised exactly by a small dynamic programme over (node, slots to
//                      spend) (Ylitie, Karras, Laine 2017, section 4.1, for 4 slots and with the reference's leaves kept).
//                      The frontier of a record is then any of the five binary-tree shapes with four leaves (or fewer slots).
// The visit order of the slots stays the reference's for every shape: depth-first over the folded BVH2 nodes, the second
// child first where the ray is negative along that node's split axis (trace_bvh.cl:181-190).

// false: the tree does not qualify (non-finite or non-nested bounds, child order): k_trace2 is used
bool build_wide_bvh(const rt_bvh_node* nodes, uint32_t nn, int collapse, std::vector<WideNode>& out, uint32_t& entry_ref,
    std::vector<uint32_t>* roots, const ownbvh::Metric* metric, const double* weights, const std::atomic<bool>* cancel, unsigned threads)
{
    auto cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
    auto is_leaf = [&](uint32_t i) { return (nodes[i].num_primitives_axis >> 16) != 0; };
    out.clear();
    if (is_leaf(0)) { entry_ref = RT_LEAF_BIT | nodes[0].offset; return true; }
    auto finite3 = [](const rt_float3& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
    // (the host's threads: an adaptation's fold -- measured weights -- is made beside the render loop on at most 16 of them)
    const unsigned T_host = threads ? threads : nn >= 262144u ? std::max(1u, std::min(std::thread::hardware_concurrency(), weights ? 16u : 32u)) : 1u;
    // fn(index) for index = 0 .. count - 1, taken by T_host threads from a shared counter
    auto on_pool = [&](size_t count, auto fn)
    {
        std::atomic<size_t> next{0};
        auto run = [&]() { for (size_t k; (k = next.fetch_add(1)) < count;) fn(k); };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T_host && t < count; ++t) pool.emplace_back(run);
        run();
        for (auto& th : pool) th.join();
    };
    // pass 0: bounds finite and exactly nested (child inside parent), children after their parent
    {
        std::atomic<bool> bad{false};
        const size_t n_slices = T_host > 1u ? 4u * T_host : 1u;
        on_pool(n_slices, [&](size_t k)
        {
            for (uint32_t i = (uint32_t)((uint64_t)nn * k / n_slices), e = (uint32_t)((uint64_t)nn * (k + 1) / n_slices); i < e; ++i)
            {
                const rt_bvh_node& n = nodes[i];
                if (!finite3(n.bounds_min) || !finite3(n.bounds_max)) { bad.store(true); return; }
                if (is_leaf(i)) continue;
                if (i + 1 >= nn || n.offset >= nn || n.offset <= i + 1 || (n.num_primitives_axis & 0xFFFFu) > 2u) { bad.store(true); return; }
                for (uint32_t c : {i + 1, n.offset})
                {
                    const rt_bvh_node& kk = nodes[c];
                    if (kk.bounds_min.x < n.bounds_min.x || kk.bounds_min.y < n.bounds_min.y || kk.bounds_min.z < n.bounds_min.z ||
                        kk.bounds_max.x > n.bounds_max.x || kk.bounds_max.y > n.bounds_max.y || kk.bounds_max.z > n.bounds_max.z)
                    { bad.store(true); return; }
                }
            }
        });
        if (bad.load()) return false;
    }
    // the collapse: split[n][k] = slots given to n's first child when n is folded with k slots to spend (k = 2..4, the second
    // child gets the rest); a child with i >= 2 slots is folded too iff open[c] has bit i set, otherwise it is one slot
    std::vector<uint8_t> split((size_t)nn * 5u, 0), open(nn, 0);
    if (collapse == RT_WIDE_SAH)
    {
        // T[n] = cost of the best collapse of n's subtree with a record rooted at n; F[n][k] = the same without the root's own
        // visit, n's subtree covered by k slots.  Children have larger indices than their parent (pass 0): one backward sweep.
        // (not value-initialised: every element that is read -- an interior node's -- is written first, by the thread that sweeps its range, which is then also
        // the thread that touches its pages)
        const std::unique_ptr<double[]> T_store(new double[nn]), F_store(new double[(size_t)nn * 5u]);
        double* const T = T_store.get();
        double* const F = F_store.get();
        auto G = [&](uint32_t c, uint32_t i) { return is_leaf(c) ? 0.0 : (i >= 2u ? std::min(T[c], F[(size_t)c * 5u + i]) : T[c]); };
        auto dp_node = [&](uint32_t n)
        {
            const uint32_t l = n + 1, r = nodes[n].offset;
            for (uint32_t k = 2; k <= 4; ++k)
            {
                double best = 0.0; uint32_t at = 0;
                for (uint32_t i = 1; i < k; ++i)
                {
                    const double c = G(l, i) + G(r, k - i);
                    if (at == 0 || c < best) { best = c; at = i; }
                }
                F[(size_t)n * 5u + k] = best;
                split[(size_t)n * 5u + k] = (uint8_t)at;
            }
            const rt_bvh_node& b = nodes[n];
            const double dx = (double)b.bounds_max.x - b.bounds_min.x, dy = (double)b.bounds_max.y - b.bounds_min.y,
                         dz = (double)b.bounds_max.z - b.bounds_min.z;
            const float bmn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, bmx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
            T[n] = (weights ? weights[n] : metric ? metric->of(bmn, bmx) : dx * dy + dy * dz + dz * dx) + F[(size_t)n * 5u + 4u];
            for (uint32_t i = 2; i <= 4; ++i)
                if (F[(size_t)n * 5u + i] < T[n]) open[n] |= (uint8_t)(1u << i);
        };
        // The sweep is backward over the array (children have larger indices).  On several threads: the array is cut into the index ranges [n, e) below
        // nodes near the root -- in the reference's depth-first layout the range of a node is its subtree, [n + 1, offset) and [offset, e) its children's --
        // every range is swept backward by one thread, then the nodes above the cut by this one.  A node whose second child lies outside its range (an array
        // that is a tree but not in that layout) makes the whole sweep start again on one thread: the ranges always partition the array, so nothing is lost.
        bool swept = false;
        if (T_host > 1u)
        {
            struct Range { uint32_t n, e; };
            std::vector<Range> ranges{{0u, nn}};
            std::vector<uint32_t> above;                                    // the nodes above the cut, parents before children
            bool layout_ok = true;
            while (ranges.size() < 16u * (size_t)T_host)
            {
                // open the largest range
                size_t at = 0;
                for (size_t i = 1; i < ranges.size(); ++i) if (ranges[i].e - ranges[i].n > ranges[at].e - ranges[at].n) at = i;
                const Range r = ranges[at];
                if (r.e - r.n < 65536u || is_leaf(r.n)) break;
                const uint32_t second = nodes[r.n].offset;
                if (second <= r.n + 1u || second >= r.e) { layout_ok = false; break; }
                above.push_back(r.n);
                ranges[at] = Range{r.n + 1u, second};
                ranges.push_back(Range{second, r.e});
            }
            if (layout_ok)
            {
                std::atomic<bool> outside{false};
                on_pool(ranges.size(), [&](size_t k)
                {
                    const Range r = ranges[k];
                    for (uint32_t n = r.e; n-- > r.n;)
                    {
                        if ((n & 0xFFFFu) == 0u && (cancelled() || outside.load(std::memory_order_relaxed))) return;
                        if (is_leaf(n)) continue;
                        if (nodes[n].offset >= r.e) { outside.store(true); return; }
                        dp_node(n);
                    }
                });
                if (cancelled()) return false;
                if (!outside.load())
                {
                    // (an opened range's node is above both of its parts: the reverse of the order of opening has children first)
                    for (size_t k = above.size(); k-- > 0;) dp_node(above[k]);
                    swept = true;
                }
                else std::fill(open.begin(), open.end(), (uint8_t)0);
            }
        }
        if (!swept)
            for (uint32_t n = nn; n-- > 0;)
            {
                if ((n & 0xFFFFu) == 0u && cancelled()) return false;
                if (is_leaf(n)) continue;
                dp_node(n);
            }
    }
    else
    {
        for (uint32_t n = 0; n < nn; ++n)
        {
            if (is_leaf(n)) continue;
            split[(size_t)n * 5u + 4u] = 2; split[(size_t)n * 5u + 3u] = 2; split[(size_t)n * 5u + 2u] = 1;
            open[n] = 1u << 2;                                              // a child with two slots to spend shows its children
        }
    }
    // One record: its slots in the BVH2's depth-first order and, per direction-sign octant, the positions in visit order.
    struct Fold { uint32_t slot[4]; uint32_t n_slots; uint8_t visit[8][4]; };
    struct Local
    {
        const rt_bvh_node* nodes; const std::vector<uint8_t>& split; const std::vector<uint8_t>& open;
        bool leaf(uint32_t i) const { return (nodes[i].num_primitives_axis >> 16) != 0; }
        // n folded with k slots to spend: appends n's slots to `f` and returns, per octant, their positions in visit order
        void fold(uint32_t n, uint32_t k, Fold& f, uint8_t (&visit)[8][4], uint32_t& count) const
        {
            const uint32_t c[2] = {n + 1, nodes[n].offset};
            const uint32_t give[2] = {split[(size_t)n * 5u + k], k - split[(size_t)n * 5u + k]};
            uint8_t part[2][8][4];
            uint32_t len[2] = {0, 0};
            for (int i = 0; i < 2; ++i)
            {
                if (!leaf(c[i]) && give[i] >= 2u && ((open[c[i]] >> give[i]) & 1u)) fold(c[i], give[i], f, part[i], len[i]);
                else
                {
                    for (int o = 0; o < 8; ++o) part[i][o][0] = (uint8_t)f.n_slots;
                    f.slot[f.n_slots++] = c[i];
                    len[i] = 1;
                }
            }
            const uint32_t axis = nodes[n].num_primitives_axis & 0xFFFFu;
            for (uint32_t o = 0; o < 8; ++o)
            {
                // trace_bvh.cl:181-190: the near child is the second one when the ray is negative along the split axis
                const int first = (int)((o >> axis) & 1u);
                uint32_t at = 0;
                for (uint32_t j = 0; j < len[first]; ++j) visit[o][at++] = part[first][o][j];
                for (uint32_t j = 0; j < len[first ^ 1]; ++j) visit[o][at++] = part[first ^ 1][o][j];
            }
            count = len[0] + len[1];
        }
        // the slots alone (pass 1 needs no visit orders)
        void slots(uint32_t n, uint32_t k, uint32_t (&slot)[4], uint32_t& count) const
        {
            const uint32_t c[2] = {n + 1, nodes[n].offset};
            const uint32_t give[2] = {split[(size_t)n * 5u + k], k - split[(size_t)n * 5u + k]};
            for (int i = 0; i < 2; ++i)
            {
                if (!leaf(c[i]) && give[i] >= 2u && ((open[c[i]] >> give[i]) & 1u)) slots(c[i], give[i], slot, count);
                else if (count < 4u) slot[count++] = c[i];
            }
        }
    } local{nodes, split, open};
    auto fold_of = [&](uint32_t n, Fold& f)
    {
        f.n_slots = 0;
        for (int k = 0; k < 4; ++k) f.slot[k] = RT_EMPTY_REF;
        uint32_t count = 0;
        local.fold(n, 4u, f, f.visit, count);
    };
    // Where the slots of a record are stored.  The kernel brings them into visit order with FOUR conditional exchanges --
    // (0,1), (2,3), (0,2), (1,3), one table bit each per direction octant: two instructions more than the three decisions
    // round 2 tabulated for the one shape it folded -- and that network does not realise every permutation; but for each of
    // the five shapes (and their smaller relatives) there is a placement of the slots for which it realises all the orders
    // the shape can ask for (exhaustive search: tests/test_wide_bvh.py).  Found here by trying the 24 placements, once per
    // distinct (slot count, eight visit orders); depth-first order is tried first, which is what the balanced shape keeps.
    struct Arrangement { uint8_t place[4]; uint32_t order; };
    typedef std::map<std::array<uint8_t, 33>, Arrangement> ArrangementCache;
    auto arrange = [&](const Fold& f, ArrangementCache& arrangements) -> const Arrangement*
    {
        std::array<uint8_t, 33> key{};
        key[0] = (uint8_t)f.n_slots;
        for (int o = 0; o < 8; ++o)
            for (uint32_t j = 0; j < f.n_slots; ++j) key[1 + 4 * o + j] = f.visit[o][j];
        auto it = arrangements.find(key);
        if (it != arrangements.end()) return &it->second;
        uint8_t place[4] = {0, 1, 2, 3};                                   // place[j] = slot position of the j-th node in depth-first order
        do
        {
            uint8_t node_at[4] = {255, 255, 255, 255};
            for (uint32_t j = 0; j < f.n_slots; ++j) node_at[place[j]] = (uint8_t)j;
            Arrangement a{};
            bool all = true;
            for (uint32_t o = 0; o < 8 && all; ++o)
            {
                bool found = false;
                for (uint32_t bits = 0; bits < 16u && !found; ++bits)
                {
                    uint8_t pos[4] = {0, 1, 2, 3};
                    static const int ex[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
                    for (int c = 0; c < 4; ++c)
                        if ((bits >> c) & 1u) std::swap(pos[ex[c][0]], pos[ex[c][1]]);
                    // the occupied slots, in the order the kernel will look at them, must be the reference's visit order
                    uint32_t at = 0;
                    bool same = true;
                    for (int k = 0; k < 4 && same; ++k)
                        if (node_at[pos[k]] != 255) same = node_at[pos[k]] == f.visit[o][at++];
                    if (same) { a.order |= bits << (4u * o); found = true; }
                }
                all = found;
            }
            if (all)
            {
                memcpy(a.place, place, 4);
                return &arrangements.emplace(key, a).first->second;
            }
        } while (std::next_permutation(place, place + 4));
        return nullptr;
    };
    // pass 1: wide nodes in depth-first order (slot 0's subtree first), like the reference's flattening.  On several threads: this one walks the records
    // down to depth 7 and notes, in the order it meets them, the records there as roots of subtrees; the pool walks those (each the same depth-first walk,
    // into a list of its own); the lists, spliced in where their roots were met, are the one-thread order.
    std::vector<uint32_t> wide_of(nn, RT_EMPTY_REF), order;
    {
        const uint32_t MARK = 0xFFFFFFFEu;                                  // reached, not numbered yet
        // a node reached twice (several parents share a child) is not a tree: the walk would append once per PATH
        auto reach = [&](uint32_t n) { return __atomic_exchange_n(&wide_of[n], MARK, __ATOMIC_RELAXED) == RT_EMPTY_REF; };
        struct Sub { uint32_t root, depth; std::vector<uint32_t> list; };
        struct Met { uint32_t what; bool sub; };
        std::vector<Sub> subs;
        std::vector<Met> met;
        std::atomic<bool> bad{false};
        // the walk from `root` (at `depth`): records to `list`; at cut_depth (0 = never) a record becomes a Sub instead of being walked
        auto walk = [&](uint32_t root, uint32_t depth0, uint32_t cut_depth, std::vector<uint32_t>* list) -> bool
        {
            std::vector<uint32_t> todo{root}, depth_of{depth0};
            size_t mine = 0;
            while (!todo.empty())
            {
                const uint32_t n = todo.back(), depth = depth_of.back();
                todo.pop_back();
                depth_of.pop_back();
                if ((mine & 0xFFFFu) == 0u && (cancelled() || bad.load(std::memory_order_relaxed))) return false;
                if (depth > 33u) return false;                                 // <= 3 pending slots per level must fit RT_W4_STACK_MAX
                if (cut_depth != 0u && depth == cut_depth) { met.push_back(Met{(uint32_t)subs.size(), true}); subs.push_back(Sub{n, depth, {}}); continue; }
                if (!reach(n) || ++mine > nn) return false;
                if (list) list->push_back(n); else met.push_back(Met{n, false});
                uint32_t slot[4] = {RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF}, count = 0;
                local.slots(n, 4u, slot, count);
                for (int k = 3; k >= 0; --k)
                    if (slot[k] != RT_EMPTY_REF && !is_leaf(slot[k])) { todo.push_back(slot[k]); depth_of.push_back(depth + 1u); }
            }
            return true;
        };
        if (T_host > 1u)
        {
            if (!walk(0u, 1u, 7u, nullptr)) return false;
            on_pool(subs.size(), [&](size_t k) { if (!walk(subs[k].root, subs[k].depth, 0u, &subs[k].list)) bad.store(true); });
            if (bad.load() || cancelled()) return false;
            size_t total = 0;
            for (const Met& m : met) total += m.sub ? subs[m.what].list.size() : 1u;
            if (total > nn) return false;
            order.reserve(total);
            std::vector<size_t> first_of(subs.size(), 0);
            for (const Met& m : met)
            {
                if (!m.sub) { wide_of[m.what] = (uint32_t)order.size(); order.push_back(m.what); continue; }
                first_of[m.what] = order.size();
                order.insert(order.end(), subs[m.what].list.begin(), subs[m.what].list.end());
            }
            on_pool(subs.size(), [&](size_t k) { for (size_t j = 0; j < subs[k].list.size(); ++j) wide_of[subs[k].list[j]] = (uint32_t)(first_of[k] + j); });
        }
        else
        {
            if (!walk(0u, 1u, 0u, &order)) return false;
            for (size_t w = 0; w < order.size(); ++w) wide_of[order[w]] = (uint32_t)w;
        }
    }
    if (order.size() >= (1u << 26)) return false;                          // 32-bit byte offsets in the kernel
    // pass 2: records (independent of each other: host threads, each with its own cache of arrangements)
    out.resize(order.size());
    auto make_record = [&](size_t w, ArrangementCache& cache) -> bool
    {
        const uint32_t n = order[w];
        Fold f;
        fold_of(n, f);
        const Arrangement* arr = arrange(f, cache);
        if (!arr) return false;                                            // cannot happen (every shape has an arrangement: tests/test_wide_bvh.py)
        uint32_t slot[4] = {RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF};
        for (uint32_t j = 0; j < f.n_slots; ++j) slot[arr->place[j]] = f.slot[j];
        WideNode& r = out[w];
        memset(&r, 0, sizeof(r));
        const float nmin[3] = {nodes[n].bounds_min.x, nodes[n].bounds_min.y, nodes[n].bounds_min.z};
        const float nmax[3] = {nodes[n].bounds_max.x, nodes[n].bounds_max.y, nodes[n].bounds_max.z};
        float origin[3];
        int exps[3];
        for (int a = 0; a < 3; ++a)
        {
            // cell = 2^e: 254 cells span the node (one spare for the floor of the origin), and the grid
            // stays exactly representable: |origin| / cell < 2^23 leaves room for + 255 below 2^24
            const double extent = (double)nmax[a] - (double)nmin[a];
            const double amax = std::max(std::fabs((double)nmin[a]), std::fabs((double)nmax[a]));
            int e = -126;
            if (extent > 0.0) e = std::max(e, (int)std::ceil(std::log2(extent / 254.0)));
            while (std::ldexp(254.0, e) < extent) ++e;
            while (amax > 0.0 && amax / std::ldexp(1.0, e) >= 8388608.0 - 256.0) ++e;
            // k_trace_w4 evaluates slab distances as q * (cell * inv) + (origin - org) * inv: bounded operands keep that
            // finite for every ray it accepts (trace_kernels.h, loop C)
            if (e > 20 || amax >= 268435456.0) return false;
            const double cell = std::ldexp(1.0, e);
            const double o = std::floor((double)nmin[a] / cell) * cell;
            origin[a] = (float)o;
            if ((double)origin[a] != o) return false;                     // cannot happen by construction
            exps[a] = e;
        }
        r.ox = origin[0]; r.oy = origin[1]; r.oz = origin[2];
        r.meta = (uint32_t)(exps[0] + 127) | (uint32_t)(exps[1] + 127) << 8 | (uint32_t)(exps[2] + 127) << 16 | f.n_slots << 24;
        r.order = arr->order;
        for (int k = 0; k < 4; ++k)
        {
            if (slot[k] == RT_EMPTY_REF)
            {
                r.ref[k] = RT_EMPTY_REF;
                for (int a = 0; a < 3; ++a) { r.lo[a] |= 255u << (8 * k); }   // lo 255 > hi 0: never hit
                continue;
            }
            const rt_bvh_node& c = nodes[slot[k]];
            r.ref[k] = is_leaf(slot[k]) ? (RT_LEAF_BIT | c.offset) : wide_of[slot[k]];
            const float cmin[3] = {c.bounds_min.x, c.bounds_min.y, c.bounds_min.z};
            const float cmax[3] = {c.bounds_max.x, c.bounds_max.y, c.bounds_max.z};
            for (int a = 0; a < 3; ++a)
            {
                const double cell = std::ldexp(1.0, exps[a]);
                double lo = std::floor(((double)cmin[a] - (double)origin[a]) / cell);
                double hi = std::ceil(((double)cmax[a] - (double)origin[a]) / cell);
                // the difference above is rounded (a bound of 1e-17 beside an origin of -0.2 vanishes in it): settle the
                // containment on the grid points themselves, which are exact in binary32 and binary64 alike
                while ((double)origin[a] + lo * cell > (double)cmin[a]) lo -= 1.0;
                while ((double)origin[a] + hi * cell < (double)cmax[a]) hi += 1.0;
                if (lo < 0.0 || hi > 255.0 || lo > hi) return false;      // cannot happen: the child is inside the node
                r.lo[a] |= (uint32_t)lo << (8 * k);
                r.hi[a] |= (uint32_t)hi << (8 * k);
            }
        }
        return true;
    };
    {
        const size_t n_records = order.size();
        // (a fold with measured weights is an adaptation's, made beside the render loop: 16 threads, adapt_threads below)
        const unsigned n_threads = (unsigned)std::min<size_t>(threads ? threads : std::max(1u, std::min(std::thread::hardware_concurrency(), weights ? 16u : 32u)), n_records / 4096 + 1);
        std::atomic<bool> ok{true};
        auto run = [&](size_t w0, size_t w1)
        {
            ArrangementCache cache;
            for (size_t w = w0; w < w1 && ok.load(std::memory_order_relaxed); ++w)
                if (((w & 0x3FFFu) == 0u && cancelled()) || !make_record(w, cache)) ok.store(false);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_records * t / n_threads, n_records * (t + 1) / n_threads);
        run(0, n_records / n_threads);
        for (auto& th : pool) th.join();
        if (!ok) return false;
    }
    entry_ref = 0;
    if (roots) *roots = order;
    return true;
}

// ---- RT_CTX_OPT_WIDE_LAYOUT = 1: the records in PAIRS (round 6) -------------------------------------------------------------------------
// The L2 of gfx950 fetches 128-byte lines (every read request of the traversal kernels at the fabric is a 128-byte one: TCC_EA0_RDREQ_128B,
// profiles/r06_fetch_size_calibration.json), so a 64-byte record that misses brings its line-mate along whether anybody wants it or not.  In the fold's
// own order -- depth first -- the line-mate of a record at an even index is its first slot's record and that of one at an odd index is whatever came
// before it.  Here the line-mate is CHOSEN: every record that has interior slots is stored at an even index with the child it hands most rays on to right
// behind it (by the weight the fold was made for: the measured crossings of an adaptation, else the area of the child's box), so a visit of that child
// never misses after the visit of its parent that must precede it.  A pure permutation of the records (refs are indices): no result depends on it.
// weight(record) -> the visit weight of the record's root box.
void pair_layout(std::vector<WideNode>& wide, std::vector<uint32_t>* roots, const std::vector<double>& weight)
{
    const uint32_t n = (uint32_t)wide.size();
    if (n < 3u) return;
    auto interior = [](uint32_t ref) { return ref != RT_EMPTY_REF && !(ref & RT_LEAF_BIT); };
    std::vector<uint32_t> order, singles, todo;
    std::vector<uint8_t> placed(n, 0);
    order.reserve(n);
    todo.push_back(0u);
    while (!todo.empty())
    {
        const uint32_t r = todo.back();
        todo.pop_back();
        if (r >= n || placed[r]) continue;
        placed[r] = 1;
        uint32_t best = RT_EMPTY_REF;
        double best_w = -1.0;
        for (uint32_t ref : wide[r].ref)
            if (interior(ref) && ref < n && !placed[ref]) { const double w = ref < weight.size() ? weight[ref] : 0.0; if (best == RT_EMPTY_REF || w > best_w) { best = ref; best_w = w; } }
        if (best == RT_EMPTY_REF) { singles.push_back(r); continue; }
        placed[best] = 1;
        order.push_back(r); order.push_back(best);
        // what hangs below the two, depth first (the head's other children before the tail's: they are the nearer relatives)
        for (int k = 3; k >= 0; --k) { const uint32_t ref = wide[best].ref[k]; if (interior(ref) && ref < n && !placed[ref]) todo.push_back(ref); }
        for (int k = 3; k >= 0; --k) { const uint32_t ref = wide[r].ref[k]; if (interior(ref) && ref < n && !placed[ref]) todo.push_back(ref); }
    }
    order.insert(order.end(), singles.begin(), singles.end());
    if (order.size() != n || order[0] != 0u) return;                       // (not a tree over all records: leave it as it is)
    std::vector<uint32_t> at(n);
    for (uint32_t i = 0; i < n; ++i) at[order[i]] = i;
    std::vector<WideNode> out(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        WideNode r = wide[order[i]];
        for (uint32_t& ref : r.ref) if (interior(ref) && ref < n) ref = at[ref];
        out[i] = r;
    }
    wide.swap(out);
    if (roots && roots->size() == n)
    {
        std::vector<uint32_t> rn(n);
        for (uint32_t i = 0; i < n; ++i) rn[i] = (*roots)[order[i]];
        roots->swap(rn);
    }
}

// the static folds' weight: the area (the own trees': their metric) of the box a record tests
void pair_layout_by_area(std::vector<WideNode>& wide, std::vector<uint32_t>& roots, const rt_bvh_node* nodes, uint32_t nn, const ownbvh::Metric* metric)
{
    if (roots.size() != wide.size()) return;
    std::vector<double> w(wide.size(), 0.0);
    for (size_t rec = 0; rec < wide.size(); ++rec)
    {
        const uint32_t node = roots[rec];
        if (node >= nn) continue;
        const rt_bvh_node& b = nodes[node];
        const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        w[rec] = metric ? metric->of(mn, mx) : dx * dy + dy * dz + dz * dx;
    }
    pair_layout(wide, &roots, w);
}

// ... an adaptation's: the measured crossings of the record's box
void pair_layout_by_node_weights(std::vector<WideNode>& wide, std::vector<uint32_t>& roots, const double* node_weights, uint32_t nn)
{
    if (roots.size() != wide.size() || !node_weights) return;
    std::vector<double> w(wide.size(), 0.0);
    for (size_t rec = 0; rec < wide.size(); ++rec) if (roots[rec] < nn) w[rec] = node_weights[roots[rec]];
    pair_layout(wide, &roots, w);
}

// The ray population a shadow tree serves (own_bvh.h): shadow rays go to the analytic lights only (hit_surface.cl:114-146,
// light.h:30-65), one uniformly chosen per hit -- towards a directional light they all share its direction, towards a point
// light they come from everywhere.
ownbvh::Metric shadow_metric(const rt_light* lights, uint32_t n, double iso_share)
{
    ownbvh::Metric m;
    m.iso = 0.0;
    for (uint32_t i = 0; i < n; ++i)
    {
        const rt_light& l = lights[i];
        const double len = std::sqrt((double)l.origin.x * l.origin.x + (double)l.origin.y * l.origin.y + (double)l.origin.z * l.origin.z);
        if (l.type == RT_LIGHT_TYPE_POINT || !(len > 0.0) || !std::isfinite(len)) { m.iso += 1.0; continue; }
        m.dirs.push_back({std::fabs(l.origin.x / len), std::fabs(l.origin.y / len), std::fabs(l.origin.z / len)});
    }
    if (m.dirs.empty()) m.iso = 1.0;
    else m.iso += iso_share * (double)m.dirs.size();   // some isotropy keeps boxes that are thin along d from growing without bound
    return m;
}

// Host threads one side of an adaptation may use beside the render loop: the closest-hit and the shadow side run together, one process per
// GPU runs one context each, so 16 + 16 threads x 8 ranks stays within a 256-core host (2 x 32 per context oversubscribed it).
unsigned adapt_threads(size_t work_items, size_t per_thread)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::min<size_t>(std::min(hw, 16u), work_items / per_thread + 1);
}
static std::atomic<uint64_t> g_truncated_walks{0};
uint64_t truncated_walks_exchange() { return g_truncated_walks.exchange(0); }
void truncated_walks_add(uint64_t n) { g_truncated_walks.fetch_add(n, std::memory_order_relaxed); }   // host walks (weights only) that met a binary tree deeper than their 126-entry stack

// counts[n] = rays whose slab test of binary-tree node n passes within [0, o.w] (plain binary32 arithmetic: a weight, not a result)
void count_box_passes(const rt_bvh_node* nodes, uint32_t nn, const float4* o, const float4* d, size_t n_rays, std::vector<uint32_t>& counts,
    const std::atomic<bool>& cancel)
{
    counts.assign(nn, 0u);
    auto run = [&](size_t r0, size_t r1)
    {
        uint32_t stack[128];
        for (size_t r = r0; r < r1 && !cancel.load(std::memory_order_relaxed); ++r)
        {
            const float org[3] = {o[r].x, o[r].y, o[r].z}, inv[3] = {1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z};
            const float t_max = o[r].w;
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0)
            {
                const uint32_t n = stack[--sp];
                const rt_bvh_node& b = nodes[n];
                const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
                float t0 = 0.0f, t1 = t_max;
                for (int a = 0; a < 3; ++a)
                {
                    const float ta = (mn[a] - org[a]) * inv[a], tb = (mx[a] - org[a]) * inv[a];
                    t0 = std::fmax(t0, std::fmin(ta, tb));         // fmin / fmax drop a NaN (0 * inf): conservative, like the kernels
                    t1 = std::fmin(t1, std::fmax(ta, tb));
                }
                if (!(t0 <= t1)) continue;
                __atomic_fetch_add(&counts[n], 1u, __ATOMIC_RELAXED);
                if ((b.num_primitives_axis >> 16) != 0) continue;
                if (sp > 125) { g_truncated_walks.fetch_add(1, std::memory_order_relaxed); continue; }
                if (b.offset >= nn || n + 1u >= nn) continue;
                stack[sp++] = b.offset;
                stack[sp++] = n + 1u;
            }
        }
    };
    const unsigned n_threads = adapt_threads(n_rays, 2048);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_rays * t / n_threads, n_rays * (t + 1) / n_threads);
    run(0, n_rays / n_threads);
    for (auto& th : pool) th.join();
}

// Mode bit 4: the ORDER in which a shadow ray looks at the slots of a record is free -- its verdict is an OR over the leaves it reaches -- and
// k_trace_w4<shadow> takes them as they are stored (the exchange network is the closest-hit rays': trace_kernels.h, w4_test_slots).  An occluded
// ray stops at its first hit, so each record's slots are stored likeliest occluder first: by how many probe shadow rays had their NEAREST occluder
// in the slot's subtree.  (tools/fold_weight_study.py --order: - 12 % steps per shadow ray on the headline scene with a quarter of them
// occluded, - 25 % for the occluded ones.)  The nearest occluder of a probe ray is found here, on the host: a plain closest-hit walk of the
// reference's binary tree with Moeller-Trumbore in binary32 -- a statistic, not a result.
void nearest_occluders(const std::vector<rt_bvh_node>& tree, const std::vector<float>& tri9, const std::vector<float4>& o, const std::vector<float4>& d,
    std::vector<uint32_t>& prim, const std::atomic<bool>& cancel)
{
    const size_t n_rays = o.size();
    const uint32_t nn = (uint32_t)tree.size(), nt = (uint32_t)(tri9.size() / 9);
    prim.assign(n_rays, RT_INVALID_ID);
    auto run = [&](size_t r0, size_t r1)
    {
        uint32_t stack[128];
        for (size_t r = r0; r < r1 && !cancel.load(std::memory_order_relaxed); ++r)
        {
            const float org[3] = {o[r].x, o[r].y, o[r].z}, dir[3] = {d[r].x, d[r].y, d[r].z}, inv[3] = {1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z};
            float t_max = o[r].w;
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0)
            {
                const uint32_t n = stack[--sp];
                const rt_bvh_node& b = tree[n];
                const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
                float t0 = 0.0f, t1 = t_max;
                for (int a = 0; a < 3; ++a)
                {
                    const float ta = (mn[a] - org[a]) * inv[a], tb = (mx[a] - org[a]) * inv[a];
                    t0 = std::fmax(t0, std::fmin(ta, tb));
                    t1 = std::fmin(t1, std::fmax(ta, tb));
                }
                if (!(t0 <= t1)) continue;
                const uint32_t count = b.num_primitives_axis >> 16;
                if (count != 0)
                {
                    for (uint32_t k = 0; k < count && b.offset + k < nt; ++k)
                    {
                        const float* p = &tri9[(size_t)(b.offset + k) * 9];
                        const float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
                        const float pv[3] = {dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0]};
                        const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                        if (!(std::fabs(det) > 1e-8f)) continue;
                        const float id = 1.0f / det;
                        const float tv[3] = {org[0] - p[0], org[1] - p[1], org[2] - p[2]};
                        const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * id;
                        if (!(u >= 0.0f && u <= 1.0f)) continue;
                        const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                        const float v = (dir[0] * qv[0] + dir[1] * qv[1] + dir[2] * qv[2]) * id;
                        if (!(v >= 0.0f && u + v <= 1.0f)) continue;
                        const float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * id;
                        if (t > 0.0f && t < t_max) { t_max = t; prim[r] = b.offset + k; }
                    }
                    continue;
                }
                if (sp > 125) { g_truncated_walks.fetch_add(1, std::memory_order_relaxed); continue; }
                if (b.offset >= nn || n + 1u >= nn) continue;
                stack[sp++] = b.offset;
                stack[sp++] = n + 1u;
            }
        }
    };
    const unsigned n_threads = adapt_threads(n_rays, 2048);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_rays * t / n_threads, n_rays * (t + 1) / n_threads);
    run(0, n_rays / n_threads);
    for (auto& th : pool) th.join();
}

// The slots of every record of `wide` (a fold of `tree`, record w testing node roots[w]) stored by descending count of probe rays whose nearest
// occluder (prim[]) lies in the slot's subtree; equal counts keep their places.  A pure permutation within each record.  Returns the records changed.
uint32_t occluder_first(std::vector<WideNode>& wide, const std::vector<uint32_t>& roots, const std::vector<rt_bvh_node>& tree, const std::vector<uint32_t>& prim)
{
    const uint32_t nn = (uint32_t)tree.size();
    if (wide.empty() || roots.size() != wide.size() || nn == 0) return 0;
    std::vector<uint32_t> parent(nn, RT_EMPTY_REF), hit(nn, 0u);
    // (the passes over the nodes and over the records in slices on host threads: every write goes to an element only one slice owns)
    const unsigned T = nn >= 262144u ? adapt_threads(nn, 131072) : 1u;
    auto in_slices = [&](size_t n, auto fn)
    {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
        fn(0u, (size_t)0, n / T);
        for (auto& th : pool) th.join();
    };
    std::vector<uint32_t> max_prim_of(T, 0u);
    in_slices(nn, [&](unsigned t, size_t b, size_t e)
    {
        uint32_t mp = 0;
        for (size_t i = b; i < e; ++i)
        {
            const uint32_t count = tree[i].num_primitives_axis >> 16;
            if (count != 0) { mp = std::max(mp, tree[i].offset + count); continue; }
            if (i + 1u < nn) parent[i + 1u] = (uint32_t)i;
            if (tree[i].offset < nn) parent[tree[i].offset] = (uint32_t)i;
        }
        max_prim_of[t] = mp;
    });
    uint32_t max_prim = 0;
    for (uint32_t mp : max_prim_of) max_prim = std::max(max_prim, mp);
    std::vector<uint32_t> leaf_of(max_prim, RT_EMPTY_REF);                 // primitive -> the leaf node of `tree` that holds it
    in_slices(nn, [&](unsigned, size_t b, size_t e)
    {
        for (size_t i = b; i < e; ++i)
        {
            const uint32_t count = tree[i].num_primitives_axis >> 16;
            for (uint32_t k = 0; k < count; ++k) leaf_of[tree[i].offset + k] = (uint32_t)i;
        }
    });
    for (uint32_t p : prim)
    {
        if (p >= max_prim) continue;
        uint32_t guard = 0;
        for (uint32_t n = leaf_of[p]; n != RT_EMPTY_REF && guard < 256u; n = parent[n], ++guard) ++hit[n];
    }
    std::vector<uint32_t> changed_of(T, 0u);
    in_slices(wide.size(), [&](unsigned t, size_t wb, size_t we)
    {
        uint32_t changed = 0;
        for (size_t w = wb; w < we; ++w)
        {
            WideNode& r = wide[w];
            uint32_t score[4]; int idx[4] = {0, 1, 2, 3};
            bool any = false;
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t ref = r.ref[k];
                uint32_t node = RT_EMPTY_REF;
                if (ref == RT_EMPTY_REF) { score[k] = 0; continue; }
                if (ref & RT_LEAF_BIT) { const uint32_t first = ref & ~RT_LEAF_BIT; node = first < max_prim ? leaf_of[first] : RT_EMPTY_REF; }
                else if (ref < roots.size()) node = roots[ref];
                score[k] = node < nn ? hit[node] + 1u : 1u;                    // occupied slots before empty ones
                any = true;
            }
            if (!any) continue;
            std::stable_sort(idx, idx + 4, [&](int x, int y) { return score[x] > score[y]; });
            if (idx[0] == 0 && idx[1] == 1 && idx[2] == 2 && idx[3] == 3) continue;
            WideNode q = r;
            for (int a = 0; a < 3; ++a) { q.lo[a] = 0; q.hi[a] = 0; }
            for (int k = 0; k < 4; ++k)
            {
                q.ref[k] = r.ref[idx[k]];
                for (int a = 0; a < 3; ++a)
                {
                    q.lo[a] |= ((r.lo[a] >> (8 * idx[k])) & 0xFFu) << (8 * k);
                    q.hi[a] |= ((r.hi[a] >> (8 * idx[k])) & 0xFFu) << (8 * k);
                }
            }
            r = q;
            ++changed;
        }
        changed_of[t] = changed;
    });
    uint32_t changed = 0;
    for (uint32_t c : changed_of) changed += c;
    return changed;
}

} // namespace rtw

// bvh.cpp -- PBRT-style binned SAH build, restated from the reference's
// src/bvh.cpp:67-245 so that, on the same input, the node array AND the
// reordered triangle array are identical (tests/test_host_pin.py checks this
// against the compiled reference):
//   * 12 buckets along the axis of maximum centroid extent (:139-150)
//   * split cost 1 + (n0*SA0 + n1*SA1)/SA (:153-169), first minimum wins (:172-181)
//   * leaf when n == 1, when all centroids coincide on that axis (:112), or when
//     n <= 4 and the best split costs at least n (:184-185)
//   * n == 2 -> median split with std::nth_element (:126-135)
//   * partition with std::partition, bucket <= best (:187-195); the relative
//     order inside each side is whatever libstdc++'s algorithm leaves, which
//     is why the same standard algorithms are used here
//   * depth-first flatten: first child = next node, second child index in
//     `offset`, leaves carry (n << 16) (:223-245)
// Differences from the reference: build nodes live in vectors instead of one `new`
// per node, recursion carries indices instead of pointers, and the build is
// PARALLEL with an identical result (the reference's single-threaded recursion
// dominates start-up for multi-million-triangle scenes):
//   * a node over more than `grain` primitives is processed by all threads:
//     bounds / centroid bounds / bucket statistics are per-chunk folds combined in
//     chunk order (min/max folds keep the first of equal values, so even the sign
//     of a zero bound matches the sequential fold), and the split reproduces
//     std::partition's arrangement exactly -- libstdc++'s bidirectional partition
//     swaps the j-th misplaced element from the left with the j-th misplaced
//     element from the right, which is computed here from per-chunk counts;
//   * the subtrees below `grain` are independent tasks built by the sequential
//     code, each into its own pool;
//   * leaves reference primitives by range start (the reordered triangle array is
//     prims order, because the depth-first build emits leaves left to right), and
//     node indices follow from subtree sizes (depth-first = pre-order).
#include "bvh.hpp"
#include <atomic>
#include <chrono>
#include <cassert>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <type_traits>
#include <functional>
#include <iostream>
#include <mutex>
#include <thread>

namespace rt
{
namespace
{
constexpr unsigned kMaxPrimsInLeaf = 4;
constexpr unsigned kBuckets = 12;

struct PrimInfo
{
    unsigned index = 0;
    Bounds3 bounds;
    float3 centroid;
};

struct BuildNode
{
    Bounds3 bounds;
    int child[2] = {-1, -1};
    int axis = 0, first = 0, count = 0;
};

// ---- a small fork-join pool -------------------------------------------------
class Pool
{
public:
    explicit Pool(unsigned n) : n_(n ? n : 1)
    {
        for (unsigned t = 1; t < n_; ++t) workers_.emplace_back([this, t] { Loop(t); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    unsigned Size() const { return n_; }
    // fn(i) for i in [0, count), dynamic distribution, returns when all are done
    void Run(unsigned count, const std::function<void(unsigned)>& fn)
    {
        if (count == 0) return;
        if (n_ == 1 || count == 1) { for (unsigned i = 0; i < count; ++i) fn(i); return; }
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; count_ = count; next_ = 0; pending_ = n_ - 1; ++epoch_; }
        cv_.notify_all();
        Work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void Work()
    {
        for (;;)
        {
            unsigned i = next_.fetch_add(1);
            if (i >= count_) break;
            (*fn_)(i);
        }
    }
    void Loop(unsigned)
    {
        unsigned seen = 0;
        for (;;)
        {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            Work();
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    unsigned n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned count_ = 0, pending_ = 0, epoch_ = 0;
    std::atomic<unsigned> next_{0};
    bool stop_ = false;
};

inline int BucketOf(const Bounds3& cb, unsigned dim, const float3& centroid)
{
    int b = kBuckets * cb.Offset(centroid)[dim];
    if (b == (int)kBuckets) b = kBuckets - 1;
    return b;
}

// split decision shared by the sequential and the parallel node code (bvh.cpp:153-185)
struct Split { bool leaf; unsigned bucket; };
inline Split ChooseSplit(const int* counts, const Bounds3* bbounds, const Bounds3& bounds, unsigned n)
{
    float cost[kBuckets - 1];
    for (unsigned i = 0; i < kBuckets - 1; ++i)
    {
        Bounds3 b0, b1;
        int c0 = 0, c1 = 0;
        for (unsigned j = 0; j <= i; ++j) { b0 = Union(b0, bbounds[j]); c0 += counts[j]; }
        for (unsigned j = i + 1; j < kBuckets; ++j) { b1 = Union(b1, bbounds[j]); c1 += counts[j]; }
        cost[i] = 1.0f + (c0 * b0.SurfaceArea() + c1 * b1.SurfaceArea()) / bounds.SurfaceArea();
    }
    float best = cost[0];
    unsigned best_bucket = 0;
    for (unsigned i = 1; i < kBuckets - 1; ++i)
        if (cost[i] < best) { best = cost[i]; best_bucket = i; }
    float leaf_cost = float(n);
    return Split{!(n > kMaxPrimsInLeaf || best < leaf_cost), best_bucket};
}

// ---- sequential build of one subtree into its own pool (ids = pre-order) -----
struct SubtreeBuilder
{
    std::vector<PrimInfo>& prims;
    std::vector<BuildNode> pool;

    int MakeLeaf(int id, unsigned start, unsigned end, const Bounds3& bounds)
    {
        pool[id].first = (int)start;     // == ordered.size() of the sequential build at this point
        pool[id].count = (int)(end - start);
        pool[id].bounds = bounds;
        return id;
    }

    int Build(unsigned start, unsigned end)
    {
        int id = (int)pool.size();
        pool.emplace_back();

        Bounds3 bounds;
        for (unsigned i = start; i < end; ++i) bounds = Union(bounds, prims[i].bounds);

        unsigned n = end - start;
        if (n == 1) return MakeLeaf(id, start, end, bounds);

        Bounds3 cb;
        for (unsigned i = start; i < end; ++i) cb = Union(cb, prims[i].centroid);
        unsigned dim = cb.MaximumExtent();
        if (cb.max[dim] == cb.min[dim]) return MakeLeaf(id, start, end, bounds);

        unsigned mid = (start + end) / 2;
        if (n <= 2)
        {
            std::nth_element(&prims[start], &prims[mid], &prims[end - 1] + 1,
                [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
        }
        else
        {
            int counts[kBuckets] = {0};
            Bounds3 bbounds[kBuckets];
            for (unsigned i = start; i < end; ++i)
            {
                int b = BucketOf(cb, dim, prims[i].centroid);
                counts[b]++;
                bbounds[b] = Union(bbounds[b], prims[i].bounds);
            }
            Split sp = ChooseSplit(counts, bbounds, bounds, n);
            if (sp.leaf) return MakeLeaf(id, start, end, bounds);
            PrimInfo* pmid = std::partition(&prims[start], &prims[end - 1] + 1, [&](const PrimInfo& pi)
            {
                return BucketOf(cb, dim, pi.centroid) <= (int)sp.bucket;
            });
            mid = (unsigned)(pmid - &prims[0]);
        }
        int c0 = Build(start, mid);
        int c1 = Build(mid, end);
        pool[id].child[0] = c0;
        pool[id].child[1] = c1;
        pool[id].bounds = Union(pool[c0].bounds, pool[c1].bounds);
        pool[id].axis = (int)dim;
        pool[id].count = 0;
        return id;
    }
};

// ---- the upper part of the tree: nodes over more than `grain` primitives ------
struct Task { unsigned start, end; SubtreeBuilder* built = nullptr; };
struct BigNode
{
    Bounds3 bounds;
    int axis = 0;
    int child[2] = {0, 0};          // >= 0: BigNode index; < 0: task ~index
    unsigned start = 0, end = 0, size = 0;
    bool leaf = false;
};

struct ParallelBuilder
{
    std::vector<PrimInfo>& prims;
    Pool& pool;
    unsigned grain;
    std::vector<BigNode> big;
    std::vector<Task> tasks;
    std::vector<unsigned char> flags;
    std::vector<unsigned> lidx, ridx;

    unsigned Chunks(unsigned n) const
    {
        unsigned c = pool.Size() * 4u;
        unsigned by_size = (n + 16383u) / 16384u;
        return std::max(1u, std::min(c, by_size));
    }

    // returns a child reference (BigNode index or ~task index)
    int Node(unsigned start, unsigned end)
    {
        unsigned n = end - start;
        if (n <= grain)
        {
            tasks.push_back(Task{start, end});
            return ~(int)(tasks.size() - 1);
        }
        int id = (int)big.size();
        big.emplace_back();
        big[id].start = start; big[id].end = end;

        const unsigned nc = Chunks(n);
        auto lo = [&](unsigned c) { return start + (unsigned)((unsigned long long)n * c / nc); };
        // per-chunk folds combined in chunk order == the sequential fold
        std::vector<Bounds3> pb(nc), pc(nc);
        pool.Run(nc, [&](unsigned c)
        {
            Bounds3 b, cb;
            for (unsigned i = lo(c); i < lo(c + 1); ++i) { b = Union(b, prims[i].bounds); cb = Union(cb, prims[i].centroid); }
            pb[c] = b; pc[c] = cb;
        });
        Bounds3 bounds, cb;
        for (unsigned c = 0; c < nc; ++c) { bounds = Union(bounds, pb[c]); cb = Union(cb, pc[c]); }
        big[id].bounds = bounds;
        unsigned dim = cb.MaximumExtent();
        if (cb.max[dim] == cb.min[dim])
        {
            big[id].leaf = true;                      // all centroids coincide (bvh.cpp:112-123)
            return id;
        }
        std::vector<int> counts(nc * kBuckets, 0);
        std::vector<Bounds3> bb(nc * kBuckets);
        pool.Run(nc, [&](unsigned c)
        {
            int* cnt = &counts[c * kBuckets];
            Bounds3* bbc = &bb[c * kBuckets];
            for (unsigned i = lo(c); i < lo(c + 1); ++i)
            {
                int b = BucketOf(cb, dim, prims[i].centroid);
                cnt[b]++;
                bbc[b] = Union(bbc[b], prims[i].bounds);
            }
        });
        int tc[kBuckets] = {0};
        Bounds3 tb[kBuckets];
        for (unsigned c = 0; c < nc; ++c)
            for (unsigned b = 0; b < kBuckets; ++b) { tc[b] += counts[c * kBuckets + b]; tb[b] = Union(tb[b], bb[c * kBuckets + b]); }
        Split sp = ChooseSplit(tc, tb, bounds, n);   // n > grain > 4: never a leaf by cost

        // std::partition(pred = bucket <= best), reproduced: the j-th element failing pred in
        // [start, mid) trades places with the j-th element passing pred, counted from the
        // right, in [mid, end)
        if (flags.size() < n) flags.resize(n);
        std::vector<unsigned> trues(nc);
        pool.Run(nc, [&](unsigned c)
        {
            unsigned t = 0;
            for (unsigned i = lo(c); i < lo(c + 1); ++i)
            {
                bool p = BucketOf(cb, dim, prims[i].centroid) <= (int)sp.bucket;
                flags[i - start] = p;
                t += p;
            }
            trues[c] = t;
        });
        unsigned total_true = 0;
        for (unsigned c = 0; c < nc; ++c) total_true += trues[c];
        const unsigned mid = start + total_true;
        std::vector<unsigned> nl(nc), nr(nc);
        pool.Run(nc, [&](unsigned c)
        {
            unsigned l = 0, r = 0;
            for (unsigned i = lo(c); i < lo(c + 1); ++i)
            {
                if (i < mid) l += !flags[i - start];
                else r += flags[i - start];
            }
            nl[c] = l; nr[c] = r;
        });
        unsigned k = 0;
        std::vector<unsigned> lbase(nc), rafter(nc);
        for (unsigned c = 0; c < nc; ++c) { lbase[c] = k; k += nl[c]; }
        unsigned acc = 0;
        for (unsigned c = nc; c-- > 0;) { rafter[c] = acc; acc += nr[c]; }
        assert(acc == k);
        if (lidx.size() < k) { lidx.resize(k); ridx.resize(k); }
        pool.Run(nc, [&](unsigned c)
        {
            unsigned l = lbase[c];
            for (unsigned i = lo(c); i < lo(c + 1) && i < mid; ++i)
                if (!flags[i - start]) lidx[l++] = i;
            unsigned r = rafter[c];                   // passing elements to the right of this chunk come first
            for (unsigned i = lo(c + 1); i-- > lo(c) && i >= mid;)
                if (flags[i - start]) ridx[r++] = i;
        });
        const unsigned sc = std::max(1u, std::min(pool.Size() * 4u, (k + 4095u) / 4096u));
        pool.Run(sc, [&](unsigned c)
        {
            unsigned a = (unsigned)((unsigned long long)k * c / sc), b = (unsigned)((unsigned long long)k * (c + 1) / sc);
            for (unsigned j = a; j < b; ++j) std::swap(prims[lidx[j]], prims[ridx[j]]);
        });

        int c0 = Node(start, mid);
        int c1 = Node(mid, end);
        big[id].child[0] = c0;
        big[id].child[1] = c1;
        big[id].axis = (int)dim;
        return id;
    }
};

unsigned EnvUnsigned(const char* name, unsigned fallback)
{
    const char* v = std::getenv(name);
    return v && *v ? (unsigned)std::strtoul(v, nullptr, 10) : fallback;
}
} // namespace

void Bvh::BuildCPU(std::vector<Triangle>& triangles)
{
    if (verbose) std::cout << "Building Bounding Volume Hierarchy for scene" << std::endl;
    nodes_.clear();
    if (triangles.empty()) return;
    auto t0 = std::chrono::steady_clock::now();
    const bool timing = std::getenv("RT_BVH_TIMING") != nullptr;
    auto lap = [&](const char* what)
    {
        if (!timing) return;
        auto t1 = std::chrono::steady_clock::now();
        std::cerr << "bvh: " << what << " " << std::chrono::duration<double>(t1 - t0).count() << " s" << std::endl;
        t0 = t1;
    };
    unsigned hw = std::thread::hardware_concurrency();
    Pool pool(EnvUnsigned("RT_BVH_THREADS", std::min(hw ? hw : 1u, 32u)));   // measured: 32 beats 64 (fork-join cost)
    const unsigned grain = std::max(8u, EnvUnsigned("RT_BVH_GRAIN", 1u << 16));
    const unsigned n = (unsigned)triangles.size();

    std::vector<PrimInfo> prims(n);
    const unsigned pc = std::max(1u, std::min(pool.Size() * 4u, (n + 16383u) / 16384u));
    auto plo = [&](unsigned c) { return (unsigned)((unsigned long long)n * c / pc); };
    pool.Run(pc, [&](unsigned c)
    {
        for (unsigned i = plo(c); i < plo(c + 1); ++i)
        {
            prims[i].index = i;
            prims[i].bounds = triangles[i].GetBounds();
            prims[i].centroid = prims[i].bounds.min * 0.5f + prims[i].bounds.max * 0.5f;
        }
    });

    lap("primitive bounds");
    // upper tree with all threads per node, then the subtrees as independent tasks
    ParallelBuilder pb{prims, pool, grain, {}, {}, {}, {}, {}};
    const int root = pb.Node(0, n);
    lap("upper tree");
    std::vector<SubtreeBuilder> subtrees;
    subtrees.reserve(pb.tasks.size());
    for (size_t t = 0; t < pb.tasks.size(); ++t) subtrees.push_back(SubtreeBuilder{prims, {}});
    pool.Run((unsigned)pb.tasks.size(), [&](unsigned t)
    {
        subtrees[t].pool.reserve((size_t)(pb.tasks[t].end - pb.tasks[t].start) * 2);
        subtrees[t].Build(pb.tasks[t].start, pb.tasks[t].end);
    });

    lap("subtrees");
    // subtree sizes (children of a BigNode were created after it), bounds of the upper nodes
    auto size_of = [&](int ref) { return ref >= 0 ? pb.big[ref].size : (unsigned)subtrees[~ref].pool.size(); };
    auto bounds_of = [&](int ref) -> const Bounds3& { return ref >= 0 ? pb.big[ref].bounds : subtrees[~ref].pool[0].bounds; };
    for (size_t i = pb.big.size(); i-- > 0;)
    {
        BigNode& b = pb.big[i];
        if (b.leaf) { b.size = 1; continue; }
        b.size = 1 + size_of(b.child[0]) + size_of(b.child[1]);
        b.bounds = Union(bounds_of(b.child[0]), bounds_of(b.child[1]));
    }

    // depth-first (pre-order) indices: node, first subtree, second subtree (bvh.cpp:223-245)
    const unsigned total = size_of(root);
    nodes_.resize(total);
    std::vector<unsigned> task_base(pb.tasks.size(), 0);
    struct Item { int ref; unsigned base; };
    std::vector<Item> todo{{root, 0}};
    while (!todo.empty())
    {
        Item it = todo.back();
        todo.pop_back();
        if (it.ref < 0) { task_base[~it.ref] = it.base; continue; }
        const BigNode& b = pb.big[it.ref];
        LinearBVHNode& out = nodes_[it.base];
        out.bounds = b.bounds;
        if (b.leaf)
        {
            assert(b.end - b.start < 65536);
            out.offset = b.start;
            out.num_primitives_axis = (std::uint32_t)(b.end - b.start) << 16;
            continue;
        }
        out.num_primitives_axis = (std::uint32_t)b.axis;
        out.offset = it.base + 1 + size_of(b.child[0]);
        todo.push_back({b.child[0], it.base + 1});
        todo.push_back({b.child[1], out.offset});
    }
    pool.Run((unsigned)pb.tasks.size(), [&](unsigned t)
    {
        const std::vector<BuildNode>& p = subtrees[t].pool;
        const unsigned base = task_base[t];
        for (size_t i = 0; i < p.size(); ++i)         // pool ids are already pre-order
        {
            LinearBVHNode& out = nodes_[base + i];
            out.bounds = p[i].bounds;
            if (p[i].count > 0)
            {
                assert(p[i].count < 65536);
                out.offset = (std::uint32_t)p[i].first;
                out.num_primitives_axis = (std::uint32_t)p[i].count << 16;
            }
            else
            {
                out.num_primitives_axis = (std::uint32_t)p[i].axis;
                out.offset = base + (std::uint32_t)p[i].child[1];
            }
        }
    });

    lap("flatten");
    // the reordered triangle array (bvh.cpp:52): leaf order == prims order.  Gathered into
    // raw storage and copied back in place, both in parallel (a value-initialised vector of
    // n Triangles would be written once more, by one thread).
    static_assert(std::is_trivially_copyable<Triangle>::value, "Triangle must be a POD record");
    std::unique_ptr<unsigned char[]> raw(new unsigned char[(size_t)n * sizeof(Triangle)]);
    Triangle* ordered = reinterpret_cast<Triangle*>(raw.get());
    pool.Run(pc, [&](unsigned c)
    {
        for (unsigned i = plo(c); i < plo(c + 1); ++i) std::memcpy(&ordered[i], &triangles[prims[i].index], sizeof(Triangle));
    });
    pool.Run(pc, [&](unsigned c)
    {
        std::memcpy(&triangles[plo(c)], &ordered[plo(c)], (size_t)(plo(c + 1) - plo(c)) * sizeof(Triangle));
    });
    lap("reorder triangles");
    if (verbose)
        std::cout << "BVH created with " << nodes_.size() << " nodes for " << triangles.size() << " triangles" << std::endl;
}
} // namespace rt

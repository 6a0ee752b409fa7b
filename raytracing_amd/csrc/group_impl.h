// group_impl.h -- the multi-GPU part of the C-ABI (include/rt_hip.h, "device groups"): tiles of one
// image rendered on several GPUs and ONE RCCL gather of the accumulated radiance to the root.
// Included at the end of rt_hip.hip (needs rt_ctx / rt_frame).
//
// The reference is single-device (src/gpu_wrappers/cl_context.cpp:89: one queue on devices_[0]); this
// is the tiling the north star adds.  Pixels are independent and the samplers are keyed by GLOBAL pixel
// coordinates, so ranks never talk while rendering; the gather is the only collective.
//
// RCCL is loaded with dlopen at the first group call instead of being linked: a process that has
// PyTorch loaded (bench.py) then shares torch's librccl.so.1 -- two copies of RCCL in one process
// would each grab the xGMI topology -- and single-GPU users never touch it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>

namespace
{
struct RcclApi
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []()
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.error = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) api.error = std::string("librccl lacks ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
        api.Gather = (decltype(api.Gather))sym("ncclGather");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return api;
}

// pixels of tile `rank` of an image cut into interleaved bands (the rule of rt_frame_create)
uint64_t tile_pixels(uint32_t width, uint32_t height, uint32_t band_h, uint32_t rank, uint32_t nranks)
{
    uint64_t rows = 0;
    for (uint64_t band = rank; band * band_h < height; band += nranks)
    {
        uint64_t start = band * band_h;
        rows += height - start < band_h ? height - start : band_h;
    }
    return rows * width;
}

// root: rank-major gathered tiles -> the row-major full image
__global__ void k_group_assemble(const float4* __restrict__ gathered, float4* __restrict__ image, uint32_t width, uint32_t height,
    uint32_t band_h, uint32_t nranks, uint64_t stride)
{
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)width * height) return;
    uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (uint64_t)y * width);
    uint32_t band = y / band_h, r = band % nranks;
    uint32_t ly = (band / nranks) * band_h + (y - band * band_h);
    image[i] = gathered[(uint64_t)r * stride + (uint64_t)ly * width + x];
}
} // namespace

struct rt_group
{
    struct Member
    {
        int device = 0, rank = 0;
        ncclComm_t comm = nullptr;
        float4* send = nullptr;        // this member's tile, padded to the largest tile
        float4* recv = nullptr;        // root only: nranks x stride
        float4* image = nullptr;       // root only: the assembled image
        size_t recv_elems = 0, image_elems = 0, send_elems = 0;
    };
    int nranks = 0;
    bool local = false;                // rt_group_create_local: all ranks on ONE device, device copies instead of RCCL
    std::vector<Member> members;       // the ranks that live in this process
    std::string error;
    // temporal denoiser across tiles (rt_group_denoise), on the root: full-image inputs and history
    float4* dn_radiance = nullptr; float* dn_depth = nullptr; float2* dn_velocity = nullptr;
    float4* dn_prev_radiance = nullptr; float* dn_prev_depth = nullptr; float4* dn_resolved = nullptr;
    size_t dn_pixels = 0;
};

namespace
{
int gfail(rt_group* g, const std::string& msg)
{
    if (g) g->error = msg;
    g_thread_error = msg;
    return RT_ERROR;
}

int ensure(rt_group* g, float4** p, size_t* have, size_t want)
{
    if (*have >= want && *p) return RT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (hipMalloc((void**)p, (want ? want : 1) * sizeof(float4)) != hipSuccess)
    {
        (void)hipGetLastError();
        return gfail(g, "rt_group: out of device memory for the gather buffers");
    }
    *have = want;
    return RT_OK;
}
} // namespace

extern "C" {

const char* rt_group_last_error(rt_group* g) { return g ? g->error.c_str() : g_thread_error.c_str(); }

static int group_create_impl(int n, const int* device_ordinals, rt_group** out, bool check_duplicates);
int rt_group_create(int n, const int* device_ordinals, rt_group** out) { return group_create_impl(n, device_ordinals, out, true); }
// The same call WITHOUT this library's own one-rank-per-device check: whatever the device list, it reaches ncclCommInitAll, and
// what RCCL answers comes back through rt_group_last_error.  For the wiring test a one-GPU box allows -- {0, 0} must fail with
// RCCL's own refusal, which proves the in-process path (`rt_render --gpus N`, TiledRender) is wired to the library up to
// communicator creation -- not for products.
int rt_group_create_unchecked(int n, const int* device_ordinals, rt_group** out) { return group_create_impl(n, device_ordinals, out, false); }

static int group_create_impl(int n, const int* device_ordinals, rt_group** out, bool check_duplicates)
{
    if (!out || n <= 0 || !device_ordinals) return gfail(nullptr, "rt_group_create: bad argument");
    *out = nullptr;
    RcclApi& api = rccl();
    if (!api.error.empty()) return gfail(nullptr, "rt_group_create: " + api.error);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return gfail(nullptr, "rt_group_create: no HIP device");
    for (int i = 0; i < n; ++i)
    {
        if (device_ordinals[i] < 0 || device_ordinals[i] >= ndev) return gfail(nullptr, "rt_group_create: bad device ordinal");
        for (int j = 0; j < i && check_duplicates; ++j)
            if (device_ordinals[j] == device_ordinals[i]) return gfail(nullptr, "rt_group_create: one rank per device (RCCL refuses two ranks on one GPU)");
    }
    std::vector<ncclComm_t> comms((size_t)n);
    ncclResult_t r = api.CommInitAll(comms.data(), n, device_ordinals);
    if (r != ncclSuccess) return gfail(nullptr, std::string("rt_group_create: ncclCommInitAll: ") + api.GetErrorString(r));
    rt_group* g = new rt_group;
    g->nranks = n;
    g->members.resize((size_t)n);
    for (int i = 0; i < n; ++i) { g->members[i].device = device_ordinals[i]; g->members[i].rank = i; g->members[i].comm = comms[i]; }
    *out = g;
    return RT_OK;
}

int rt_group_create_local(int n, int device_ordinal, rt_group** out)
{
    if (!out || n <= 0) return gfail(nullptr, "rt_group_create_local: bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return gfail(nullptr, "rt_group_create_local: no HIP device");
    if (device_ordinal < 0 || device_ordinal >= ndev) return gfail(nullptr, "rt_group_create_local: bad device ordinal");
    rt_group* g = new rt_group;
    g->nranks = n;
    g->local = true;
    g->members.resize((size_t)n);
    for (int i = 0; i < n; ++i) { g->members[i].device = device_ordinal; g->members[i].rank = i; }
    *out = g;
    return RT_OK;
}

int rt_group_unique_id(void* id_bytes, size_t capacity)
{
    if (!id_bytes || capacity < sizeof(ncclUniqueId)) return gfail(nullptr, "rt_group_unique_id: buffer smaller than RT_GROUP_ID_BYTES");
    RcclApi& api = rccl();
    if (!api.error.empty()) return gfail(nullptr, "rt_group_unique_id: " + api.error);
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return gfail(nullptr, std::string("rt_group_unique_id: ") + api.GetErrorString(r));
    memcpy(id_bytes, &id, sizeof(id));
    return RT_OK;
}

int rt_group_join(int nranks, int rank, const void* id_bytes, int device_ordinal, rt_group** out)
{
    if (!out || !id_bytes || nranks <= 0 || rank < 0 || rank >= nranks) return gfail(nullptr, "rt_group_join: bad argument");
    *out = nullptr;
    RcclApi& api = rccl();
    if (!api.error.empty()) return gfail(nullptr, "rt_group_join: " + api.error);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return gfail(nullptr, "rt_group_join: no HIP device");
    if (device_ordinal < 0 || device_ordinal >= ndev) return gfail(nullptr, "rt_group_join: bad device ordinal");
    if (hipSetDevice(device_ordinal) != hipSuccess) return gfail(nullptr, "rt_group_join: hipSetDevice failed");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = api.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return gfail(nullptr, std::string("rt_group_join: ncclCommInitRank: ") + api.GetErrorString(r));
    rt_group* g = new rt_group;
    g->nranks = nranks;
    g->members.resize(1);
    g->members[0].device = device_ordinal;
    g->members[0].rank = rank;
    g->members[0].comm = comm;
    *out = g;
    return RT_OK;
}

int rt_group_size(rt_group* g) { return g ? g->nranks : 0; }

// What RCCL itself says about the communicator of local member i: ncclCommCount (ranks) and ncclCommUserRank.  This is the
// proof that the collective spans the ranks the caller believes it does (rt_group_size only echoes the caller's argument).
// A local group (rt_group_create_local: device copies, no RCCL) reports 0 ranks.
int rt_group_comm_count(rt_group* g, int i, int* comm_ranks, int* comm_user_rank)
{
    if (!g || i < 0 || i >= (int)g->members.size()) return gfail(g, "rt_group_comm_count: bad argument");
    if (comm_ranks) *comm_ranks = 0;
    if (comm_user_rank) *comm_user_rank = -1;
    if (g->local) return RT_OK;
    RcclApi& api = rccl();
    if (!api.error.empty()) return gfail(g, "rt_group_comm_count: " + api.error);
    rt_group::Member& m = g->members[(size_t)i];
    int n = 0, r = -1;
    ncclResult_t e = api.CommCount(m.comm, &n);
    if (e == ncclSuccess) e = api.CommUserRank(m.comm, &r);
    if (e != ncclSuccess) return gfail(g, std::string("rt_group_comm_count: ") + api.GetErrorString(e));
    if (comm_ranks) *comm_ranks = n;
    if (comm_user_rank) *comm_user_rank = r;
    return RT_OK;
}
int rt_group_local_count(rt_group* g) { return g ? (int)g->members.size() : 0; }
int rt_group_local_rank(rt_group* g, int i) { return g && i >= 0 && i < (int)g->members.size() ? g->members[(size_t)i].rank : -1; }

} // extern "C"

namespace
{
// checks that frames[i] is the tile of local member i of ONE image; returns that image's geometry and the
// padded tile size (pixels of the largest tile)
int check_tiles(rt_group* g, rt_frame* const* frames, const char* who, uint32_t& width, uint32_t& height, uint32_t& band_h, uint64_t& stride)
{
    const rt_frame* f0 = frames[0];
    if (!f0) return gfail(g, std::string(who) + ": NULL frame");
    width = f0->tile.width; height = f0->tile.height; band_h = f0->tile.band_h;
    stride = 0;
    for (int r = 0; r < g->nranks; ++r)
    {
        uint64_t p = tile_pixels(width, height, band_h, (uint32_t)r, (uint32_t)g->nranks);
        stride = p > stride ? p : stride;
    }
    for (size_t i = 0; i < g->members.size(); ++i)
    {
        rt_frame* f = frames[i];
        rt_group::Member& m = g->members[i];
        if (!f) return gfail(g, std::string(who) + ": NULL frame");
        if (f->ctx->device != m.device) return gfail(g, std::string(who) + ": frame lives on another device than its rank");
        if (f->tile.width != width || f->tile.height != height || f->tile.band_h != band_h || (int)f->tile.nranks != g->nranks ||
            (int)f->tile.rank != m.rank)
            return gfail(g, std::string(who) + ": frame is not the tile of this rank (rt_frame_desc tile_rank / tile_count / band_height)");
    }
    return RT_OK;
}

// THE collective: every member's `send` (floats_per_rank floats) lands in the root's `recv` at rank * floats_per_rank.
// RCCL ncclGather on the frames' streams; a local group (one device) copies instead.
int exchange(rt_group* g, rt_frame* const* frames, int root, size_t floats_per_rank, const char* who)
{
    const size_t nm = g->members.size();
    if (g->local)
    {
        rt_group::Member* rootm = nullptr;
        size_t root_i = 0;
        for (size_t i = 0; i < nm; ++i) if (g->members[i].rank == root) { rootm = &g->members[i]; root_i = i; }
        if (!rootm) return gfail(g, std::string(who) + ": root is not in this process");
        for (size_t i = 0; i < nm; ++i)
            if (hipStreamSynchronize(frames[i]->ctx->stream) != hipSuccess) return gfail(g, std::string(who) + ": stream synchronisation failed");
        for (size_t i = 0; i < nm; ++i)
        {
            hipError_t e = hipMemcpyAsync((float*)rootm->recv + (size_t)g->members[i].rank * floats_per_rank, g->members[i].send,
                floats_per_rank * sizeof(float), hipMemcpyDeviceToDevice, frames[root_i]->ctx->stream);
            if (e != hipSuccess) return gfail(g, std::string(who) + ": device copy: " + hipGetErrorString(e));
        }
        return RT_OK;
    }
    RcclApi& api = rccl();
    ncclResult_t r = api.GroupStart();
    for (size_t i = 0; i < nm && r == ncclSuccess; ++i)
    {
        rt_group::Member& m = g->members[i];
        (void)hipSetDevice(m.device);
        r = api.Gather(m.send, m.rank == root ? m.recv : nullptr, floats_per_rank, ncclFloat, root, m.comm, frames[i]->ctx->stream);
    }
    ncclResult_t r2 = api.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return gfail(g, std::string(who) + ": ncclGather: " + api.GetErrorString(r));
    return RT_OK;
}

// root: one section (COMPS floats per pixel, rank-major padded tiles) of the gathered buffer -> row-major image
template <int COMPS>
__global__ void k_group_assemble_n(const float* __restrict__ gathered, float* __restrict__ image, uint32_t width, uint32_t height,
    uint32_t band_h, uint32_t nranks, uint64_t floats_per_rank, uint64_t section_offset)
{
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)width * height) return;
    uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (uint64_t)y * width);
    uint32_t band = y / band_h, r = band % nranks;
    uint32_t ly = (band / nranks) * band_h + (y - band * band_h);
    const float* src = gathered + (uint64_t)r * floats_per_rank + section_offset + ((uint64_t)ly * width + x) * COMPS;
    for (int c = 0; c < COMPS; ++c) image[i * COMPS + c] = src[c];
}
} // namespace

extern "C" {

int rt_group_gather_radiance(rt_group* g, rt_frame* const* frames, int root, float* host_rgba, void** device_rgba)
{
    if (!g || !frames) return gfail(g, "rt_group_gather_radiance: NULL argument");
    if (root < 0 || root >= g->nranks) return gfail(g, "rt_group_gather_radiance: bad root");
    if (device_rgba) *device_rgba = nullptr;
    const size_t nm = g->members.size();
    uint32_t width, height, band_h;
    uint64_t stride;
    if (check_tiles(g, frames, "rt_group_gather_radiance", width, height, band_h, stride) != RT_OK) return RT_ERROR;
    // stage: running-sum radiance of every local tile into its padded send buffer (stream-ordered after the render)
    for (size_t i = 0; i < nm; ++i)
    {
        rt_frame* f = frames[i];
        rt_group::Member& m = g->members[i];
        if (hipSetDevice(m.device) != hipSuccess) return gfail(g, "rt_group_gather_radiance: hipSetDevice failed");
        if (ensure(g, &m.send, &m.send_elems, (size_t)stride * 2) != RT_OK) return RT_ERROR;      // x2: room for rt_group_denoise's 7 floats
        if (m.rank == root)
        {
            if (ensure(g, &m.recv, &m.recv_elems, (size_t)stride * 2 * (size_t)g->nranks) != RT_OK) return RT_ERROR;
            if (ensure(g, &m.image, &m.image_elems, (size_t)width * height) != RT_OK) return RT_ERROR;
        }
        if (flush_log_keep(f) != RT_OK) return gfail(g, std::string("rt_group_gather_radiance: ") + f->ctx->error);
        if (f->n_local)
        {
            hipError_t e = hipMemcpyAsync(m.send, f->radiance, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToDevice, f->ctx->stream);
            if (e != hipSuccess) return gfail(g, std::string("rt_group_gather_radiance: staging copy: ") + hipGetErrorString(e));
        }
    }
    if (exchange(g, frames, root, (size_t)stride * 4, "rt_group_gather_radiance") != RT_OK) return RT_ERROR;
    // root: un-interleave the bands, hand the image over
    for (size_t i = 0; i < nm; ++i)
    {
        rt_group::Member& m = g->members[i];
        hipStream_t s = frames[i]->ctx->stream;
        (void)hipSetDevice(m.device);
        if (m.rank == root)
        {
            uint64_t n = (uint64_t)width * height;
            hipLaunchKernelGGL(k_group_assemble, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, s, (const float4*)m.recv, m.image,
                width, height, band_h, (uint32_t)g->nranks, stride);
            if (hipGetLastError() != hipSuccess) return gfail(g, "rt_group_gather_radiance: assemble kernel launch failed");
            if (host_rgba)
            {
                hipError_t e = hipMemcpyAsync(host_rgba, m.image, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) return gfail(g, std::string("rt_group_gather_radiance: read-back: ") + hipGetErrorString(e));
            }
            if (device_rgba) *device_rgba = m.image;
        }
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return gfail(g, std::string("rt_group_gather_radiance: ") + hipGetErrorString(e));
    }
    return RT_OK;
}

// Temporal denoiser across tiles (denoiser.cl:27-79 reprojects across rows, so a tile cannot run it alone):
// gather-then-denoise on the root.  Every rank has rendered ONE sample of its tile with RT_OPT_DENOISER on
// (reset + sample + AOVs, integrator.cpp:29-46); one gather carries radiance, depth and motion vectors
// (7 floats per pixel); the root runs TemporalAccumulation on the assembled frame against ITS history, copies
// the history (cl_pt_integrator.cpp:670-675) and resolves (resolve_radiance.cl with ENABLE_DENOISER).
int rt_group_denoise(rt_group* g, rt_frame* const* frames, int root, float* host_resolved_rgba, float* host_radiance_rgba)
{
    if (!g || !frames) return gfail(g, "rt_group_denoise: NULL argument");
    if (root < 0 || root >= g->nranks) return gfail(g, "rt_group_denoise: bad root");
    const size_t nm = g->members.size();
    uint32_t width, height, band_h;
    uint64_t stride;
    if (check_tiles(g, frames, "rt_group_denoise", width, height, band_h, stride) != RT_OK) return RT_ERROR;
    const size_t per_rank = (size_t)stride * 7;                          // radiance 4 | depth 1 | velocity 2
    for (size_t i = 0; i < nm; ++i)
    {
        rt_frame* f = frames[i];
        rt_group::Member& m = g->members[i];
        if (f->denoiser != 2) return gfail(g, "rt_group_denoise: the frames must run with RT_OPT_DENOISER = 2 (inputs only)");
        if (hipSetDevice(m.device) != hipSuccess) return gfail(g, "rt_group_denoise: hipSetDevice failed");
        if (ensure(g, &m.send, &m.send_elems, (size_t)stride * 2) != RT_OK) return RT_ERROR;
        if (m.rank == root && ensure(g, &m.recv, &m.recv_elems, (size_t)stride * 2 * (size_t)g->nranks) != RT_OK) return RT_ERROR;
        if (flush_log_keep(f) != RT_OK) return gfail(g, std::string("rt_group_denoise: ") + f->ctx->error);
        if (f->n_local)
        {
            float* send = (float*)m.send;
            hipStream_t s = f->ctx->stream;
            bool ok = hipMemcpyAsync(send, f->radiance, (size_t)f->n_local * 16, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(send + (size_t)stride * 4, f->aov_buf.depth, (size_t)f->n_local * 4, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(send + (size_t)stride * 5, f->aov_buf.velocity, (size_t)f->n_local * 8, hipMemcpyDeviceToDevice, s) == hipSuccess;
            if (!ok) return gfail(g, "rt_group_denoise: staging copy failed");
        }
    }
    if (exchange(g, frames, root, per_rank, "rt_group_denoise") != RT_OK) return RT_ERROR;
    for (size_t i = 0; i < nm; ++i)
    {
        rt_group::Member& m = g->members[i];
        hipStream_t s = frames[i]->ctx->stream;
        (void)hipSetDevice(m.device);
        if (m.rank == root)
        {
            const uint64_t n = (uint64_t)width * height;
            if (g->dn_pixels != n)
            {
                auto free_dn = [&]()
                {
                    for (void** p : {(void**)&g->dn_radiance, (void**)&g->dn_depth, (void**)&g->dn_velocity, (void**)&g->dn_prev_radiance,
                             (void**)&g->dn_prev_depth, (void**)&g->dn_resolved})
                    {
                        if (*p) (void)hipFree(*p);
                        *p = nullptr;                                       // never freed twice, whatever fails below
                    }
                    g->dn_pixels = 0;
                };
                free_dn();
                bool ok = hipMalloc((void**)&g->dn_radiance, n * 16) == hipSuccess && hipMalloc((void**)&g->dn_depth, n * 4) == hipSuccess &&
                          hipMalloc((void**)&g->dn_velocity, n * 8) == hipSuccess && hipMalloc((void**)&g->dn_prev_radiance, n * 16) == hipSuccess &&
                          hipMalloc((void**)&g->dn_prev_depth, n * 4) == hipSuccess && hipMalloc((void**)&g->dn_resolved, n * 16) == hipSuccess;
                ok = ok && hipMemsetAsync(g->dn_prev_radiance, 0, n * 16, s) == hipSuccess && hipMemsetAsync(g->dn_prev_depth, 0, n * 4, s) == hipSuccess;
                if (!ok) { free_dn(); (void)hipGetLastError(); return gfail(g, "rt_group_denoise: out of device memory"); }
                g->dn_pixels = n;
            }
            const dim3 grid((uint32_t)((n + 255u) / 256u)), block(256);
            const float* recv = (const float*)m.recv;
            hipLaunchKernelGGL((k_group_assemble_n<4>), grid, block, 0, s, recv, (float*)g->dn_radiance, width, height, band_h,
                (uint32_t)g->nranks, (uint64_t)per_rank, (uint64_t)0);
            hipLaunchKernelGGL((k_group_assemble_n<1>), grid, block, 0, s, recv, g->dn_depth, width, height, band_h,
                (uint32_t)g->nranks, (uint64_t)per_rank, (uint64_t)stride * 4);
            hipLaunchKernelGGL((k_group_assemble_n<2>), grid, block, 0, s, recv, (float*)g->dn_velocity, width, height, band_h,
                (uint32_t)g->nranks, (uint64_t)per_rank, (uint64_t)stride * 5);
            hipLaunchKernelGGL(k_denoise, grid, block, 0, s, width, height, g->dn_radiance, (const float4*)g->dn_prev_radiance,
                (const float*)g->dn_depth, (const float*)g->dn_prev_depth, (const float2*)g->dn_velocity);
            bool ok = hipGetLastError() == hipSuccess &&
                      hipMemcpyAsync(g->dn_prev_radiance, g->dn_radiance, n * 16, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(g->dn_prev_depth, g->dn_depth, n * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
            DAov none = {nullptr, nullptr, nullptr, nullptr};
            hipLaunchKernelGGL(k_resolve, grid, block, 0, s, (const float4*)g->dn_radiance, none, g->dn_resolved, (uint32_t)n, 1u, 0u, 1u);
            ok = ok && hipGetLastError() == hipSuccess;
            if (ok && host_resolved_rgba) ok = hipMemcpyAsync(host_resolved_rgba, g->dn_resolved, n * 16, hipMemcpyDeviceToHost, s) == hipSuccess;
            if (ok && host_radiance_rgba) ok = hipMemcpyAsync(host_radiance_rgba, g->dn_radiance, n * 16, hipMemcpyDeviceToHost, s) == hipSuccess;
            if (!ok) return gfail(g, "rt_group_denoise: denoise / resolve on the root failed");
        }
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return gfail(g, std::string("rt_group_denoise: ") + hipGetErrorString(e));
    }
    return RT_OK;
}

int rt_group_destroy(rt_group* g)
{
    if (!g) return RT_OK;
    for (auto& m : g->members)
    {
        (void)hipSetDevice(m.device);
        for (float4* p : {m.send, m.recv, m.image}) if (p) (void)hipFree(p);
        if (m.comm && rccl().CommDestroy) (void)rccl().CommDestroy(m.comm);
    }
    for (void* p : {(void*)g->dn_radiance, (void*)g->dn_depth, (void*)g->dn_velocity, (void*)g->dn_prev_radiance, (void*)g->dn_prev_depth,
             (void*)g->dn_resolved})
        if (p) (void)hipFree(p);
    delete g;
    return RT_OK;
}

} // extern "C"

// main.cpp -- headless command line front end with the reference's flags
// (src/main.cpp:34-58: -w -h --scene --scale --flip_yz) plus the knobs the GUI
// exposed (--bounces, --furnace, aperture/focus) and --spp / --out for batch use.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include "render.hpp"

static void WritePFM(const char* path, const std::vector<float>& rgba, unsigned w, unsigned h)
{
    FILE* f = fopen(path, "wb");
    if (!f) return;
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for (unsigned y = h; y-- > 0;)            // PFM stores the bottom row first
        for (unsigned x = 0; x < w; ++x) fwrite(&rgba[((size_t)y * w + x) * 4], sizeof(float), 3, f);
    fclose(f);
}

int main(int argc, char** argv)
{
    try
    {
        unsigned width = 1280, height = 720, spp = 16, bounces = 3, gpus = 1, frames = 0, samples_ahead = 1;
        std::string scene_path = "assets/ShaderBalls.obj", out, save_cache;
        float scale = 1.0f, aperture = 0.0f, focus = 10.0f;
        bool flip_yz = false, furnace = false, tiled_path = false, shared_device = false, plan_only = false, resolve = true;
        unsigned scene_options = 0;      // rt::Scene::Options (opt-in extensions)
        for (int i = 1; i < argc; ++i)
        {
            auto next = [&]() -> const char* { if (i + 1 >= argc) { std::cerr << "missing value for " << argv[i] << "\n"; exit(2); } return argv[++i]; };
            if (!strcmp(argv[i], "-w")) width = (unsigned)atoi(next());
            else if (!strcmp(argv[i], "-h")) height = (unsigned)atoi(next());
            else if (!strcmp(argv[i], "--scene")) scene_path = next();
            else if (!strcmp(argv[i], "--scale")) scale = (float)atof(next());
            else if (!strcmp(argv[i], "--flip_yz")) flip_yz = atoi(next()) != 0;
            else if (!strcmp(argv[i], "--spp")) spp = (unsigned)atoi(next());
            else if (!strcmp(argv[i], "--bounces")) bounces = (unsigned)atoi(next());
            else if (!strcmp(argv[i], "--furnace")) furnace = atoi(next()) != 0;
            else if (!strcmp(argv[i], "--aperture")) aperture = (float)atof(next());
            else if (!strcmp(argv[i], "--focus")) focus = (float)atof(next());
            else if (!strcmp(argv[i], "--out")) out = next();
            else if (!strcmp(argv[i], "--save-cache")) save_cache = next();
            else if (!strcmp(argv[i], "--gpus")) gpus = (unsigned)atoi(next());
            else if (!strcmp(argv[i], "--resolve")) resolve = atoi(next()) != 0;          // with --frames: 0 = no ResolveRadiance / present per frame (diagnostic)
            else if (!strcmp(argv[i], "--frames")) frames = (unsigned)atoi(next());      // the reference's interactive loop, headless: n x RenderFrame()
            else if (!strcmp(argv[i], "--samples_ahead")) samples_ahead = (unsigned)atoi(next());   // with --frames: RT_OPT_SAMPLES_AHEAD (1 = the integrator's default, 0 = off)
            else if (!strcmp(argv[i], "--tiled")) tiled_path = atoi(next()) != 0;      // take the TiledRender path even with one GPU
            else if (!strcmp(argv[i], "--shared_device")) shared_device = atoi(next()) != 0;   // all tiles on GPU 0 (device copies instead of RCCL)
            else if (!strcmp(argv[i], "--plan")) plan_only = atoi(next()) != 0;              // print the tiling and exit: no GPU, no scene
            else if (!strcmp(argv[i], "--wide_texture_indices")) { if (atoi(next()) != 0) scene_options |= rt::Scene::kWideTextureIndices; }
            else if (!strcmp(argv[i], "--emissive_nee")) { if (atoi(next()) != 0) scene_options |= rt::Scene::kEmissiveNee; }
            else if (!strcmp(argv[i], "--help"))
            {
                std::cout << "rt_render -w W -h H --scene file.obj [--scale s] [--flip_yz 0|1] [--spp n] [--bounces b]"
                             " [--furnace 0|1] [--aperture a] [--focus d] [--out image.pfm] [--save-cache scene.rtscene] [--gpus n]\n"
                             "  --gpus n tiles the image over devices 0..n-1 (interleaved 8-row bands, one RCCL gather);\n"
                             "  --shared_device 1 puts all n tiles on GPU 0 (device copies instead of RCCL); --plan 1 prints the tiling and exits\n"
                             "  --scene also accepts a file written by --save-cache (parsed scene + BVH)\n"
                             "  --frames n times the reference's own loop instead of a batch: n x Render::RenderFrame() = one Integrate() through\n"
                             "  the fifteen hooks, one sample per pixel, ResolveRadiance + Finish() every frame (src/render.cpp:172-204);\n"
                             "  --samples_ahead 0 makes every one of them trace its own sample (default 1: a standing camera's next samples are traced ahead in batches)\n"
                             "  extensions (off = the reference's behaviour): --wide_texture_indices 1 loads scenes with more than 255\n"
                             "  textures; --emissive_nee 1 adds the emissive triangles to next-event estimation\n";
                return 0;
            }
        }
        if (plan_only)
        {
            // which rows each GPU renders (TiledRender::TileRows = rt_frame_desc's rule); needs neither a GPU nor the scene
            std::size_t total = 0;
            for (unsigned r = 0; r < gpus; ++r)
            {
                std::vector<std::uint32_t> rows = rt::TiledRender::TileRows(height, r, gpus);
                total += rows.size();
                std::cout << "tile " << r << " of " << gpus << ": " << rows.size() << " rows x " << width << " =";
                for (std::uint32_t y : rows) std::cout << " " << y;
                std::cout << std::endl;
            }
            std::cout << "total rows " << total << " of " << height << std::endl;
            return total == height ? 0 : 1;
        }
        rt::Scene scene(scene_path.c_str(), scale, flip_yz, scene_options);
        scene.AddDirectionalLight({-0.6f, -1.5f, 3.5f}, {15.0f, 10.0f, 5.0f});   // main.cpp:58
        if (gpus > 1 || tiled_path)
        {
            std::vector<int> devices;
            for (unsigned d = 0; d < gpus; ++d) devices.push_back(shared_device ? 0 : (int)d);
            rt::TiledRender tiled(width, height, scene, devices);
            for (unsigned d = 0; d < gpus; ++d) std::cout << "tile " << d << " on device " << devices[d] << ": " << tiled.GetContext(d).DeviceName() << std::endl;
            std::cout << "gather: " << (tiled.GetRcclRanks() ? "RCCL ncclGather, communicator of " + std::to_string(tiled.GetRcclRanks()) + " ranks"
                                                              : std::string("device copies on one GPU (local group)")) << std::endl;
            if (!save_cache.empty()) scene.SaveCache(save_cache.c_str(), tiled.GetAccelerationStructure().GetNodes());
            rt::Camera cam = rt::DefaultCamera(width, height);
            cam.aperture = aperture;
            cam.focus_distance = focus;
            tiled.SetCamera(cam);
            tiled.SetMaxBounces(bounces);
            tiled.EnableWhiteFurnace(furnace);
            auto t0 = std::chrono::steady_clock::now();
            tiled.RenderSamples(spp);
            double t_render = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<float> sum = tiled.GatherRadiance(0);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            rt_stats st = tiled.GetStats();
            double rays = (double)st.closest_rays + (double)st.shadow_rays;
            std::cout << spp << " spp on " << gpus << " GPUs in " << dt << " s (render " << t_render << " s, gather "
                      << (dt - t_render) * 1e3 << " ms), " << rays / dt / 1e6 << " Mrays/s; tile seconds:";
            for (double t : tiled.GetLastTileSeconds()) std::cout << " " << t;
            std::cout << std::endl;
            if (!out.empty())
            {
                for (float& v : sum) v /= (float)spp;
                WritePFM(out.c_str(), sum, width, height);
            }
            return 0;
        }
        rt::Render render(width, height, scene);
        std::cout << "device: " << render.GetContext().DeviceName() << std::endl;
        if (!save_cache.empty()) scene.SaveCache(save_cache.c_str(), render.GetAccelerationStructure().GetNodes());
        rt::Camera cam = rt::DefaultCamera(width, height);
        cam.aperture = aperture;
        cam.focus_distance = focus;
        render.SetCamera(cam);
        render.GetIntegrator().SetMaxBounces(bounces);
        render.GetIntegrator().EnableWhiteFurnace(furnace);
        if (frames != 0)
        {
            // timing run: wait for the fold adaptation instead of adopting it whenever its worker is done (RT_CTX_OPT_ADAPTIVE_FOLD | 2, as bench.py does)
            if (rt_ctx_set_option(render.GetContext().Get(), RT_CTX_OPT_ADAPTIVE_FOLD, 27u) != RT_OK) throw rt::HIPException("rt_ctx_set_option failed");
            render.UploadGPUData();
            // The reference's main loop without its window (src/main.cpp:62-72 -> Render::RenderFrame, src/render.cpp:172-204): every frame is one
            // Integrate() through the hooks and ends with ResolveRadiance + Finish().  A warm-up batch first (the fold adaptation happens there),
            // then `frames` timed frames.  No Python, no PyTorch in this process: the HIP runtime is the system's.
            render.GetIntegrator().SetResolveEveryFrame(resolve);
            render.GetIntegrator().SetSamplesAhead(samples_ahead);
            render.RenderSamples(8);
            for (int i = 0; i < 24; ++i) render.RenderFrame();             // (the backend times its two ways over a scene's first 20 frames: RT_OPT_FRAME_KERNEL = 255)
            render.GetContext().Finish();
            rt_stats s0 = render.GetIntegrator().GetStats();
            auto tf = std::chrono::steady_clock::now();
            for (unsigned i = 0; i < frames; ++i) render.RenderFrame();
            (void)render.GetIntegrator().GetResolvedImage();               // the last image has arrived
            render.GetContext().Finish();
            double df = std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();
            rt_stats s1 = render.GetIntegrator().GetStats();
            double frays = (double)(s1.closest_rays - s0.closest_rays) + (double)(s1.shadow_rays - s0.shadow_rays);
            // (RT_OPT_SAMPLES_AHEAD: the ray totals include the samples traced ahead at either end -- scaled to the timed frames' share; bench.py counts exactly)
            const double traced = (double)frames + (double)s1.samples_ahead - (double)s0.samples_ahead;
            if (traced > 0.0) frays *= (double)frames / traced;
            std::cout << frames << " frames (one Integrate() each, resolve + Finish() every frame) in " << df << " s: " << df * 1e3 / frames
                      << " ms per frame, " << frays / df / 1e6 << " Mrays/s (" << (s1.samples_from_banks - s0.samples_from_banks) << " of them replayed from batches traced ahead)" << std::endl;
            return 0;
        }
        auto t0 = std::chrono::steady_clock::now();
        render.RenderSamples(spp);
        render.GetContext().Finish();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rt_stats st = render.GetIntegrator().GetStats();
        double rays = (double)st.closest_rays + (double)st.shadow_rays;
        std::cout << spp << " spp in " << dt << " s, " << rays / dt / 1e6 << " Mrays/s" << std::endl;
        if (!out.empty())
        {
            std::vector<float> sum = render.GetIntegrator().ReadRadianceSum();
            for (float& v : sum) v /= (float)spp;
            WritePFM(out.c_str(), sum, width, height);
        }
    }
    catch (std::exception& ex)
    {
        std::cerr << "Caught exception: " << ex.what() << std::endl;   // main.cpp:74-77
        return 1;
    }
    return 0;
}

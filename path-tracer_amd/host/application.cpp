// application.cpp -- see application.hpp (source/application.cpp:13-90, source/main.cpp:3-7).
#include "application.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

Application::Application(int argc, char** argv)
{
    _info.width = 1920; _info.height = 1080;
    _info.accumMode = PT_ACCUM_SUM_F32;
    bool sceneGiven = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--scene")) { if (!sceneGiven) _info.scene.clear(); sceneGiven = true; _info.scene.push_back(next()); }
        else if (!std::strcmp(argv[i], "--width")) _info.width = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--height")) _info.height = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--spp")) _info.samplesPerFrame = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--depth")) _info.maxDepth = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--frames")) _frames = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--device")) _device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus")) _gpus = static_cast<uint32_t>(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--rehearse")) _rehearse = true;
        else if (!std::strcmp(argv[i], "--dump-accum")) _dumpAccum = next();
        else if (!std::strcmp(argv[i], "--literal-8bit")) _info.accumMode = PT_ACCUM_MEAN_UNORM8;
        else if (!std::strcmp(argv[i], "--clock-seed")) _info.seedMode = PT_SEED_CLOCK;
        else if (!std::strcmp(argv[i], "--eye")) { for (float& v : _info.eye) v = static_cast<float>(std::atof(next())); }
        else if (!std::strcmp(argv[i], "--center")) { for (float& v : _info.center) v = static_cast<float>(std::atof(next())); }
        else if (!std::strcmp(argv[i], "--fov")) _info.fovDegrees = static_cast<float>(std::atof(next()));
        else if (!std::strcmp(argv[i], "--output")) _output = next();
        else if (!std::strcmp(argv[i], "--dump-result")) _dumpResult = next();
        else { std::fprintf(stderr, "usage: pt_render [--scene file.gltf|file.ptscene]... [--width W --height H] [--spp S] [--depth D] [--frames N]\n"
                                    "                 [--eye x y z] [--center x y z] [--fov deg] [--literal-8bit] [--clock-seed] [--device N]\n"
                                    "                 [--gpus N [--rehearse]] [--output image.ppm] [--dump-result frame.f32] [--dump-accum sum.f32]\n"); std::exit(2); }
    }
    if (_gpus < 1 || _gpus > 16) { std::fprintf(stderr, "--gpus must be in [1, 16]\n"); std::exit(2); }
    if (_gpus > 1 && _info.accumMode != PT_ACCUM_SUM_F32) { std::fprintf(stderr, "--gpus needs the float sum image (the 8-bit running mean is not associative)\n"); std::exit(2); }
    for (uint32_t r = 0; r < _gpus; ++r) {
        // scene, BVH and tables are replicated: one context and one renderer per GPU
        _contexts.push_back(std::make_shared<HipContext>(_gpus == 1 || _rehearse ? _device : static_cast<int>(r)));
        _renderers.push_back(std::make_unique<Renderer>(_info, _contexts.back()));
    }
}

int Application::Run()
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < _frames; ++f) _renderers[f % _gpus]->Render(f);
    if (_gpus > 1) {
        std::vector<PtContext*> ctxs; std::vector<void*> images;
        for (const auto& r : _renderers) { ctxs.push_back(r->Context()->Handle()); images.push_back(r->RenderTarget()); }
        PtCheckResult(pt_accum_allreduce(ctxs.data(), images.data(), _gpus, _info.width * _info.height), "Failed summing the images!");
    }
    PtTraceStats stats {};
    for (const auto& r : _renderers) {                 // waits for every stream
        const PtTraceStats s = r->Stats();
        stats.rays += s.rays; stats.paths += s.paths;
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!_dumpResult.empty()) {
        const std::vector<float> r = _renderers[(_frames - 1) % _gpus]->LastFrameResult();
        FILE* f = std::fopen(_dumpResult.c_str(), "wb");
        if (!f || std::fwrite(r.data(), sizeof(float), r.size(), f) != r.size()) { std::fprintf(stderr, "[FILE] cannot write %s\n", _dumpResult.c_str()); return 1; }
        std::fclose(f);
    }
    if (!_dumpAccum.empty()) {
        const std::vector<float> a = _renderers.back()->ReadAccumulation();     // every renderer holds the total
        FILE* f = std::fopen(_dumpAccum.c_str(), "wb");
        if (!f || std::fwrite(a.data(), sizeof(float), a.size(), f) != a.size()) { std::fprintf(stderr, "[FILE] cannot write %s\n", _dumpAccum.c_str()); return 1; }
        std::fclose(f);
    }
    if (!_output.empty()) {
        const std::vector<uint8_t> img = _renderers[0]->ReadImageRGBA8();
        FILE* f = std::fopen(_output.c_str(), "wb");
        if (!f) { std::fprintf(stderr, "[FILE] cannot write %s\n", _output.c_str()); return 1; }
        std::fprintf(f, "P6\n%u %u\n255\n", _info.width, _info.height);
        for (size_t i = 0; i < img.size(); i += 4) std::fwrite(&img[i], 1, 3, f);
        std::fclose(f);
    }
    std::printf("{\"frames\": %u, \"gpus\": %u, \"width\": %u, \"height\": %u, \"spp_per_frame\": %u, \"rays\": %llu, \"paths\": %llu, \"seconds\": %.6f, \"rays_per_s\": %.1f, \"paths_per_s\": %.1f}\n",
                _frames, _gpus, _info.width, _info.height, _info.samplesPerFrame, static_cast<unsigned long long>(stats.rays), static_cast<unsigned long long>(stats.paths),
                seconds, stats.rays / seconds, stats.paths / seconds);
    return 0;
}

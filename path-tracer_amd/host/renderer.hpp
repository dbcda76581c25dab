// renderer.hpp -- mirrors include/renderer.hpp:20-89 of the reference without the presentation half:
// the constructor loads the scene, builds one BLAS per (node, mesh) and the TLAS, uploads the
// bindless tables and the camera; Render() traces one frame into the accumulation target.
#pragma once
#include "acceleration_structure.hpp"

struct RendererInitInfo {       // stands where the reference passes VulkanInitInfo (width, height, surface)
    uint32_t width = 0, height = 0;
    std::vector<std::string> scene = { "assets/cornell/CornellBox-Original.gltf" };   // source/renderer.cpp:26-28
    uint32_t samplesPerFrame = 25;   // shaders/ray_gen.rgen:34
    uint32_t maxDepth = 10;          // shaders/ray_gen.rgen:52
    uint32_t accumMode = PT_ACCUM_MEAN_UNORM8;   // the reference's 8-bit render target
    uint32_t seedMode = PT_SEED_SAMPLE_INDEX;
    float eye[3] = { 0.0f, 1.0f, 3.0f }, center[3] = { 0.0f, 1.0f, 0.0f };   // source/renderer.cpp:182
    float fovDegrees = 60.0f;
};

class Renderer {
public:
    Renderer(const RendererInitInfo& initInfo, const std::shared_ptr<HipContext>& context);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void Render();
    // Multi-GPU: renderer r of N traces frames r, r + N, ... of the same image, so the frame index is the caller's.
    void Render(uint32_t frameIndex);

    // headless additions: results come back through these instead of a swapchain
    [[nodiscard]] std::vector<float> LastFrameResult() const;        // width*height*3, value of `result` in ray_gen.rgen:74
    [[nodiscard]] std::vector<uint8_t> ReadImageRGBA8() const;       // resolved accumulation target
    [[nodiscard]] PtTraceStats Stats() const;
    [[nodiscard]] uint32_t RenderedFrames() const { return _renderedFrames; }
    [[nodiscard]] void* RenderTarget() const { return _renderTarget; }           // device pointer of the accumulation image
    [[nodiscard]] std::vector<float> ReadAccumulation() const;                   // float sum / mean image, width*height*4
    [[nodiscard]] const std::shared_ptr<HipContext>& Context() const { return _context; }

private:
    struct CameraUniformData { float viewInverse[16]; float projInverse[16]; };   // include/renderer.hpp:31-35
    void InitializeRenderTarget();
    void InitializeCamera();
    void InitializeBLAS();

    std::shared_ptr<HipContext> _context;
    RendererInitInfo _info;
    void* _renderTarget = nullptr;
    uint32_t _renderedFrames = 0;
    std::unique_ptr<ModelLoader> _modelLoader;
    std::shared_ptr<BindlessResources> _bindlessResources;
    std::vector<std::shared_ptr<Model>> _models {};
    std::vector<BottomLevelAccelerationStructure> _blases {};
    std::unique_ptr<TopLevelAccelerationStructure> _tlas;
    CameraUniformData _cameraData {};
    uint32_t _windowWidth = 0, _windowHeight = 0;
};

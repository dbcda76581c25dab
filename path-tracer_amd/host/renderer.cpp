// renderer.cpp -- source/renderer.cpp of the reference restated over the C ABI:
// ctor :13-42, Render :60-107 + RecordCommands :109-122, InitializeRenderTarget :161-170,
// InitializeCamera :172-193, InitializeBLASInput :417-455, InitializeBLAS :457-470.
#include "renderer.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>

Renderer::Renderer(const RendererInitInfo& initInfo, const std::shared_ptr<HipContext>& context)
    : _context(context), _info(initInfo), _windowWidth(initInfo.width), _windowHeight(initInfo.height)
{
    InitializeRenderTarget();
    _bindlessResources = std::make_shared<BindlessResources>(_context);
    _modelLoader = std::make_unique<ModelLoader>(_bindlessResources, _context);
    for (const auto& modelPath : _info.scene) {
        std::shared_ptr<Model> model = _modelLoader->LoadFromFile(modelPath);
        if (!model) { std::fprintf(stderr, "[FILE] Aborting: model %s could not be loaded\n", modelPath.c_str()); std::abort(); }
        _models.push_back(model);
    }
    InitializeBLAS();
    _tlas = std::make_unique<TopLevelAccelerationStructure>(_blases, _bindlessResources, _context);
    _bindlessResources->UpdateDescriptorSet();
    InitializeCamera();
}

Renderer::~Renderer()
{
    _tlas.reset();
    _blases.clear();
    if (_renderTarget) _context->Free(_renderTarget);
}

void Renderer::Render()
{
    PtCheckResult(pt_trace(_context->Handle(), _tlas->Structure(), _cameraData.viewInverse, _windowWidth, _windowHeight, _renderedFrames,
                           _info.samplesPerFrame, _info.maxDepth, _info.seedMode, _info.accumMode, _renderTarget),
                  "Failed tracing rays!");
    _renderedFrames++;
}

void Renderer::Render(uint32_t frameIndex)
{
    PtCheckResult(pt_trace(_context->Handle(), _tlas->Structure(), _cameraData.viewInverse, _windowWidth, _windowHeight, frameIndex,
                           _info.samplesPerFrame, _info.maxDepth, _info.seedMode, _info.accumMode, _renderTarget),
                  "Failed tracing rays!");
    _renderedFrames++;
}

std::vector<float> Renderer::ReadAccumulation() const
{
    std::vector<float> out(size_t(_windowWidth) * _windowHeight * 4);
    if (_info.accumMode != PT_ACCUM_MEAN_UNORM8) _context->Download(_renderTarget, out.size() * sizeof(float), out.data());
    return out;
}

void Renderer::InitializeRenderTarget()
{
    const size_t pixels = size_t(_windowWidth) * _windowHeight;
    const size_t bytes = pixels * (_info.accumMode == PT_ACCUM_MEAN_UNORM8 ? 4 : 16);
    std::vector<uint8_t> zeros(bytes, 0);
    _renderTarget = _context->Upload(zeros.data(), bytes);
}

// glm::perspectiveRH_ZO(radians(fov), aspect, 0.1, 1000) with [1][1] negated, inverted in closed form,
// and the inverse of lookAt(eye, center, (0,1,0)); both stored column-major like glm::mat4.
void Renderer::InitializeCamera()
{
    const float aspect = _windowWidth / static_cast<float>(_windowHeight);
    const float t = static_cast<float>(std::tan(double(_info.fovDegrees) * 0.017453292519943295 / 2.0));
    const float zNear = 0.1f, zFar = 1000.0f;
    const float p00 = 1.0f / (aspect * t), p11 = -(1.0f / t), p22 = zFar / (zNear - zFar), p32 = -(zFar * zNear) / (zFar - zNear);
    float* PI = _cameraData.projInverse;   // PI[c*4 + r]
    for (int i = 0; i < 16; ++i) PI[i] = 0.0f;
    PI[0 * 4 + 0] = static_cast<float>(1.0 / double(p00));
    PI[1 * 4 + 1] = static_cast<float>(1.0 / double(p11));
    PI[3 * 4 + 2] = -1.0f;                                        // z_out = -w_in
    PI[2 * 4 + 3] = static_cast<float>(1.0 / double(p32));        // w_out = z_in / p32 + w_in * p22 / p32
    PI[3 * 4 + 3] = static_cast<float>(double(p22) / double(p32));
    double f[3], s[3], u[3];
    double len = 0;
    for (int k = 0; k < 3; ++k) { f[k] = double(_info.center[k]) - _info.eye[k]; len += f[k] * f[k]; }
    len = std::sqrt(len);
    for (int k = 0; k < 3; ++k) f[k] /= len;
    const double up[3] = { 0, 1, 0 };
    s[0] = f[1] * up[2] - f[2] * up[1]; s[1] = f[2] * up[0] - f[0] * up[2]; s[2] = f[0] * up[1] - f[1] * up[0];
    len = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (int k = 0; k < 3; ++k) s[k] /= len;
    u[0] = s[1] * f[2] - s[2] * f[1]; u[1] = s[2] * f[0] - s[0] * f[2]; u[2] = s[0] * f[1] - s[1] * f[0];
    float* VI = _cameraData.viewInverse;   // inverse of a rigid transform: columns s, u, -f, eye
    for (int i = 0; i < 16; ++i) VI[i] = 0.0f;
    for (int r = 0; r < 3; ++r) {
        VI[0 * 4 + r] = static_cast<float>(s[r]);
        VI[1 * 4 + r] = static_cast<float>(u[r]);
        VI[2 * 4 + r] = static_cast<float>(-f[r]);
        VI[3 * 4 + r] = _info.eye[r];
    }
    VI[15] = 1.0f;
}

static BLASInput InitializeBLASInput(const std::shared_ptr<Model>& model, const Node& node, const Mesh& mesh)
{
    BLASInput output {};
    output.transform = node.GetWorldMatrix();
    output.vertexData = reinterpret_cast<uint64_t>(model->vertexBuffer->devicePtr);
    output.indexData = reinterpret_cast<uint64_t>(model->indexBuffer->devicePtr) + mesh.firstIndex * sizeof(uint32_t);
    output.maxVertex = model->verticesCount - 1;
    output.vertexStride = sizeof(Model::Vertex);
    output.primitiveCount = mesh.indexCount / 3;
    output.node.vertexBufferDeviceAddress = output.vertexData;
    output.node.indexBufferDeviceAddress = output.indexData;
    output.node.material = mesh.material;
    return output;
}

void Renderer::InitializeBLAS()
{
    for (const auto& model : _models)
        for (const auto& node : model->nodes)
            for (const auto mesh : node.meshes) {
                BLASInput input = InitializeBLASInput(model, node, model->meshes[mesh]);
                _blases.emplace_back(input, _bindlessResources, _context);
            }
}

std::vector<float> Renderer::LastFrameResult() const
{
    std::vector<float> out(size_t(_windowWidth) * _windowHeight * 3);
    PtCheckResult(pt_trace_last_result(_context->Handle(), out.data()), "Failed reading the frame result!");
    return out;
}

std::vector<uint8_t> Renderer::ReadImageRGBA8() const
{
    const size_t pixels = size_t(_windowWidth) * _windowHeight;
    std::vector<uint8_t> out(pixels * 4);
    if (_info.accumMode == PT_ACCUM_MEAN_UNORM8) { _context->Download(_renderTarget, out.size(), out.data()); return out; }
    std::vector<uint8_t> zeros(pixels * 4, 0);
    void* rgba = _context->Upload(zeros.data(), zeros.size());
    // float sum: rgb / alpha; float running mean: alpha is 1, so the same resolve divides by one
    PtCheckResult(pt_accum_resolve(_context->Handle(), static_cast<const float*>(_renderTarget), static_cast<uint32_t>(pixels), static_cast<uint8_t*>(rgba), 0), "Failed resolving!");
    _context->Download(rgba, out.size(), out.data());
    _context->Free(rgba);
    return out;
}

PtTraceStats Renderer::Stats() const
{
    PtTraceStats s {};
    PtCheckResult(pt_trace_stats(_context->Handle(), &s), "Failed reading stats!");
    return s;
}

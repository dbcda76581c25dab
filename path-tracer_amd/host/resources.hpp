// resources.hpp -- host-side resource tables of the path tracer.
//
// Mirrors include/resources/resource_manager.hpp:6-36 (ResourceHandle / ResourceManager),
// include/resources/gpu_resources.hpp:112-205 (MaterialCreation, Material, GeometryNode, BLASInstance)
// and include/resources/bindless_resources.hpp:43-93 (BindlessResources) of the reference, with the
// Vulkan buffers, images and descriptor sets replaced by device allocations made through the C ABI.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"

constexpr uint32_t NULL_RESOURCE_INDEX_VALUE = 0xFFFF;

template <typename T>
struct ResourceHandle {
    bool operator==(const ResourceHandle<T>& other) const { return handle == other.handle; }
    static ResourceHandle<T> Null() { return ResourceHandle<T> {}; }
    [[nodiscard]] bool IsNull() const { return handle == NULL_RESOURCE_INDEX_VALUE; }
    uint32_t handle = NULL_RESOURCE_INDEX_VALUE;
};

template <typename T>
class ResourceManager {
public:
    const T& Get(ResourceHandle<T> handle) const { return _resources[handle.handle]; }
    const std::vector<T>& GetAll() const { return _resources; }

protected:
    ResourceHandle<T> Create(T&& resource)
    {
        uint32_t index = static_cast<uint32_t>(_resources.size());
        _resources.push_back(std::move(resource));
        return ResourceHandle<T> { index };
    }

private:
    std::vector<T> _resources {};
};

// Aborts with the library's message on a non-zero status, like VkCheckResult (source/vk_common.cpp:5-15).
void PtCheckResult(int status, const char* what);

// Device + stream; stands where the reference passes std::shared_ptr<VulkanContext>.
class HipContext {
public:
    explicit HipContext(int device = 0);
    ~HipContext();
    HipContext(const HipContext&) = delete;
    HipContext& operator=(const HipContext&) = delete;
    [[nodiscard]] PtContext* Handle() const { return _ctx; }
    void* Upload(const void* host, size_t bytes) const;
    void Download(const void* dev, size_t bytes, void* host) const;
    void Free(void* dev) const;

private:
    PtContext* _ctx = nullptr;
};

struct Buffer {   // device allocation owned by the host object that created it
    Buffer(const void* host, size_t bytes, const std::shared_ptr<HipContext>& context);
    ~Buffer();
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    void* devicePtr = nullptr;
    size_t size = 0;

private:
    std::shared_ptr<HipContext> _context;
};

struct ImageCreation {
    std::vector<uint8_t> data {};   // RGBA8, top row first
    uint32_t width {}, height {};
    std::string name {};
};

struct Image {
    Image(const ImageCreation& creation, const std::shared_ptr<HipContext>& context);
    std::shared_ptr<Buffer> texels;
    uint32_t width {}, height {};
};

struct Material;
struct MaterialCreation {   // defaults of include/resources/gpu_resources.hpp:112-133
    ResourceHandle<Image> albedoMap = ResourceHandle<Image>::Null();
    std::array<float, 4> albedoFactor { 1.0f, 1.0f, 1.0f, 1.0f };
    ResourceHandle<Image> metallicRoughnessMap = ResourceHandle<Image>::Null();
    float metallicFactor = 0.0f;
    float roughnessFactor = 0.0f;
    ResourceHandle<Image> normalMap = ResourceHandle<Image>::Null();
    float normalScale = 0.0f;
    ResourceHandle<Image> occlusionMap = ResourceHandle<Image>::Null();
    float occlusionStrength = 0.0f;
    ResourceHandle<Image> emissiveMap = ResourceHandle<Image>::Null();
    std::array<float, 3> emissiveFactor { 0.0f, 0.0f, 0.0f };
};

struct Material : PtMaterial {   // 96 bytes, the std140 layout of shaders/bindless.glsl:3-27
    explicit Material(const MaterialCreation& creation);
};
static_assert(sizeof(Material) == 96, "Material must match the device layout");

struct GeometryNodeCreation {
    uint64_t vertexBufferDeviceAddress = 0;
    uint64_t indexBufferDeviceAddress = 0;
    ResourceHandle<Material> material = ResourceHandle<Material>::Null();
};
struct GeometryNode : PtGeometryNode {
    explicit GeometryNode(const GeometryNodeCreation& creation);
};
static_assert(sizeof(GeometryNode) == 32, "GeometryNode must match the device layout");

using BLASInstance = PtBLASInstance;
using BLASInstanceCreation = BLASInstance;

class ImageResources : public ResourceManager<Image> {
public:
    explicit ImageResources(const std::shared_ptr<HipContext>& context) : _context(context) {}
    ResourceHandle<Image> Create(const ImageCreation& creation) { return ResourceManager<Image>::Create(Image(creation, _context)); }

private:
    std::shared_ptr<HipContext> _context;
};
class MaterialResources : public ResourceManager<Material> {
public:
    ResourceHandle<Material> Create(const MaterialCreation& creation) { return ResourceManager<Material>::Create(Material(creation)); }
    ResourceHandle<Material> CreateRaw(const PtMaterial& raw);
};
class GeometryNodeResources : public ResourceManager<GeometryNode> {
public:
    ResourceHandle<GeometryNode> Create(const GeometryNodeCreation& creation) { return ResourceManager<GeometryNode>::Create(GeometryNode(creation)); }
};
class BLASInstanceResources : public ResourceManager<BLASInstance> {
public:
    ResourceHandle<BLASInstance> Create(const BLASInstanceCreation& creation) { return ResourceManager<BLASInstance>::Create(BLASInstance(creation)); }
};

class BindlessResources {
public:
    explicit BindlessResources(const std::shared_ptr<HipContext>& context);
    void UpdateDescriptorSet();   // uploads the four tables: pt_scene_set_tables
    [[nodiscard]] ImageResources& Images() { return _imageResources; }
    [[nodiscard]] MaterialResources& Materials() { return _materialResources; }
    [[nodiscard]] GeometryNodeResources& GeometryNodes() { return _geometryNodeResources; }
    [[nodiscard]] BLASInstanceResources& BLASInstances() { return _blasInstanceResources; }

private:
    static constexpr uint32_t MAX_RESOURCES = 1024;
    std::shared_ptr<HipContext> _context;
    ImageResources _imageResources;
    MaterialResources _materialResources;
    GeometryNodeResources _geometryNodeResources {};
    BLASInstanceResources _blasInstanceResources {};
    ResourceHandle<Image> _fallbackImage;
};

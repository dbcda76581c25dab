// resources.cpp -- see resources.hpp.  Table packing follows source/resources/gpu_resources.cpp:414-441,
// table upload source/resources/bindless_resources.cpp:70-233.
#include "resources.hpp"
#include <cstdio>
#include <cstdlib>

void PtCheckResult(int status, const char* what)
{
    if (status == 0) return;
    std::fprintf(stderr, "[HIP] %s: %s\n", what, pt_last_error());
    std::abort();
}

HipContext::HipContext(int device) { PtCheckResult(pt_ctx_create(device, nullptr, &_ctx), "Failed creating context!"); }
HipContext::~HipContext() { pt_ctx_destroy(_ctx); }
void* HipContext::Upload(const void* host, size_t bytes) const
{
    void* d = nullptr;
    PtCheckResult(pt_buffer_upload(_ctx, host, bytes, &d), "Failed uploading buffer!");
    return d;
}
void HipContext::Download(const void* dev, size_t bytes, void* host) const { PtCheckResult(pt_buffer_download(_ctx, dev, bytes, host), "Failed downloading buffer!"); }
void HipContext::Free(void* dev) const { pt_buffer_free(_ctx, dev); }

Buffer::Buffer(const void* host, size_t bytes, const std::shared_ptr<HipContext>& context)
    : size(bytes), _context(context)
{
    devicePtr = context->Upload(host, bytes);
}
Buffer::~Buffer() { if (devicePtr) _context->Free(devicePtr); }

Image::Image(const ImageCreation& creation, const std::shared_ptr<HipContext>& context)
    : width(creation.width), height(creation.height)
{
    texels = std::make_shared<Buffer>(creation.data.data(), creation.data.size(), context);
}

Material::Material(const MaterialCreation& c)
{
    std::memset(static_cast<PtMaterial*>(this), 0, sizeof(PtMaterial));
    useAlbedoMap = !c.albedoMap.IsNull();
    useMRMap = !c.metallicRoughnessMap.IsNull();
    useNormalMap = !c.normalMap.IsNull();
    useOcclusionMap = !c.occlusionMap.IsNull();
    useEmissiveMap = !c.emissiveMap.IsNull();
    albedoMapIndex = c.albedoMap.handle;
    mrMapIndex = c.metallicRoughnessMap.handle;
    normalMapIndex = c.normalMap.handle;
    occlusionMapIndex = c.occlusionMap.handle;
    emissiveMapIndex = c.emissiveMap.handle;
    for (int k = 0; k < 4; ++k) albedoFactor[k] = c.albedoFactor[k];
    metallicFactor = c.metallicFactor;
    roughnessFactor = c.roughnessFactor;
    normalScale = c.normalScale;
    occlusionStrength = c.occlusionStrength;
    for (int k = 0; k < 3; ++k) emissiveFactor[k] = c.emissiveFactor[k];
}

ResourceHandle<Material> MaterialResources::CreateRaw(const PtMaterial& raw)
{
    Material m { MaterialCreation {} };
    static_cast<PtMaterial&>(m) = raw;
    return ResourceManager<Material>::Create(std::move(m));
}

GeometryNode::GeometryNode(const GeometryNodeCreation& c)
{
    std::memset(static_cast<PtGeometryNode*>(this), 0, sizeof(PtGeometryNode));
    vertexBufferDeviceAddress = c.vertexBufferDeviceAddress;
    indexBufferDeviceAddress = c.indexBufferDeviceAddress;
    materialIndex = c.material.handle;
}

BindlessResources::BindlessResources(const std::shared_ptr<HipContext>& context)
    : _context(context), _imageResources(context)
{
    // 2x2 all-zero fallback image occupies index 0 (source/resources/bindless_resources.cpp:52-61)
    ImageCreation fallback {};
    fallback.name = "Fallback texture";
    fallback.width = fallback.height = 2;
    fallback.data.assign(2 * 2 * 4, 0);
    _fallbackImage = _imageResources.Create(fallback);
}

void BindlessResources::UpdateDescriptorSet()
{
    const auto& images = _imageResources.GetAll();
    std::vector<PtTextureDesc> textures(images.size());
    for (size_t i = 0; i < images.size(); ++i)
        textures[i] = PtTextureDesc { reinterpret_cast<uint64_t>(images[i].texels->devicePtr), images[i].width, images[i].height };
    const auto& materials = _materialResources.GetAll();
    const auto& nodes = _geometryNodeResources.GetAll();
    const auto& instances = _blasInstanceResources.GetAll();
    std::vector<PtMaterial> m(materials.begin(), materials.end());
    std::vector<PtGeometryNode> g(nodes.begin(), nodes.end());
    PtCheckResult(pt_scene_set_tables(_context->Handle(), m.data(), static_cast<uint32_t>(m.size()), g.data(), static_cast<uint32_t>(g.size()),
                                      instances.data(), static_cast<uint32_t>(instances.size()), textures.data(), static_cast<uint32_t>(textures.size())),
                  "Failed uploading bindless tables!");
}

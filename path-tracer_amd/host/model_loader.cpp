// model_loader.cpp -- see model_loader.hpp.  Follows source/model_loader.cpp of the reference:
// ProcessMaterial :57-126, ProcessMesh :128-191, ProcessNode(s) :193-251, GetWorldMatrix :253-265,
// LoadFromFile :273-288, ProcessModel :290-362.
#include "model_loader.hpp"
#include "json.hpp"
#include <cmath>
#include <cstdio>
#include <fstream>
#include <functional>
#include <sstream>

Mat4 Mat4::Identity()
{
    Mat4 r {};
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
    return r;
}

Mat4 operator*(const Mat4& a, const Mat4& b)
{
    Mat4 r {};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

Mat4 Node::GetWorldMatrix() const
{
    Mat4 matrix = localMatrix;
    const Node* p = parent;
    while (p) {
        matrix = p->localMatrix * matrix;
        p = p->parent;
    }
    return matrix;
}

ModelLoader::ModelLoader(const std::shared_ptr<BindlessResources>& bindlessResources, const std::shared_ptr<HipContext>& context)
    : _context(context), _bindlessResources(bindlessResources)
{
}

std::shared_ptr<Model> ModelLoader::LoadFromFile(std::string_view path)
{
    std::fprintf(stderr, "[FILE] Loading model file %.*s\n", static_cast<int>(path.size()), path.data());
    const std::string p(path);
    _imageCache.clear();
    std::shared_ptr<Model> model;
    try {
        if (p.size() > 8 && p.compare(p.size() - 8, 8, ".ptscene") == 0) model = LoadPtScene(p);
        else model = LoadGltf(p);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "[FILE] Failed to load model file %s with error: %s\n", p.c_str(), e.what());
        return nullptr;
    }
    return model;
}

void ModelLoader::Upload(Model& model)
{
    model.verticesCount = static_cast<uint32_t>(model.hostVertices.size());
    model.indexCount = static_cast<uint32_t>(model.hostIndices.size());
    model.vertexBuffer = std::make_unique<Buffer>(model.hostVertices.data(), sizeof(Model::Vertex) * model.hostVertices.size(), _context);
    model.indexBuffer = std::make_unique<Buffer>(model.hostIndices.data(), sizeof(uint32_t) * model.hostIndices.size(), _context);
}

// ------------------------------------------------------------------------------------------- glTF
namespace {

std::string ReadFile(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string Base64Decode(const std::string& in)
{
    std::string out;
    uint32_t acc = 0; int bits = 0;
    for (char c : in) {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52; else if (c == '+') v = 62; else if (c == '/') v = 63; else continue;
        acc = (acc << 6) | static_cast<uint32_t>(v); bits += 6;
        if (bits >= 8) { bits -= 8; out += static_cast<char>((acc >> bits) & 0xFF); }
    }
    return out;
}

struct Accessor {
    const uint8_t* base; size_t stride; size_t count; int componentType; int components;
    double get(size_t i, int c) const
    {
        const uint8_t* p = base + i * stride;
        switch (componentType) {
        case 5120: return reinterpret_cast<const int8_t*>(p)[c];
        case 5121: return p[c];
        case 5122: { int16_t v; std::memcpy(&v, p + 2 * c, 2); return v; }
        case 5123: { uint16_t v; std::memcpy(&v, p + 2 * c, 2); return v; }
        case 5125: { uint32_t v; std::memcpy(&v, p + 4 * c, 4); return v; }
        default: { float v; std::memcpy(&v, p + 4 * c, 4); return v; }
        }
    }
    float getf(size_t i, int c) const { float v; std::memcpy(&v, base + i * stride + 4 * c, 4); return v; }
};

int ComponentSize(int t) { return t == 5120 || t == 5121 ? 1 : (t == 5122 || t == 5123 ? 2 : 4); }
int ComponentCount(const std::string& t) { return t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 16; }

// aiNode::mTransformation as Assimp's glTF2 importer composes it: T * R * S in binary32
Mat4 NodeMatrix(const Json& n)
{
    Mat4 m = Mat4::Identity();
    if (n.has("matrix")) {
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m.m[r][c] = static_cast<float>(n["matrix"][size_t(c * 4 + r)].num);
        return m;
    }
    if (n.has("translation")) {
        Mat4 t = Mat4::Identity();
        for (int k = 0; k < 3; ++k) t.m[k][3] = static_cast<float>(n["translation"][size_t(k)].num);
        m = m * t;
    }
    if (n.has("rotation")) {
        const float x = static_cast<float>(n["rotation"][size_t(0)].num), y = static_cast<float>(n["rotation"][size_t(1)].num);
        const float z = static_cast<float>(n["rotation"][size_t(2)].num), w = static_cast<float>(n["rotation"][size_t(3)].num);
        Mat4 r = Mat4::Identity();
        r.m[0][0] = 1.0f - 2.0f * (y * y + z * z); r.m[0][1] = 2.0f * (x * y - z * w); r.m[0][2] = 2.0f * (x * z + y * w);
        r.m[1][0] = 2.0f * (x * y + z * w); r.m[1][1] = 1.0f - 2.0f * (x * x + z * z); r.m[1][2] = 2.0f * (y * z - x * w);
        r.m[2][0] = 2.0f * (x * z - y * w); r.m[2][1] = 2.0f * (y * z + x * w); r.m[2][2] = 1.0f - 2.0f * (x * x + y * y);
        m = m * r;
    }
    if (n.has("scale")) {
        Mat4 s = Mat4::Identity();
        for (int k = 0; k < 3; ++k) s.m[k][k] = static_cast<float>(n["scale"][size_t(k)].num);
        m = m * s;
    }
    return m;
}

} // namespace

std::shared_ptr<Model> ModelLoader::LoadGltf(const std::string& path)
{
    const std::string directory = path.substr(0, path.find_last_of('/'));
    const Json j = Json::parse(ReadFile(path));
    std::vector<std::string> buffers;
    for (const Json& b : j["buffers"].arr) {
        const std::string& uri = b["uri"].str;
        if (uri.rfind("data:", 0) == 0) buffers.push_back(Base64Decode(uri.substr(uri.find(',') + 1)));
        else buffers.push_back(ReadFile(directory + "/" + uri));
    }
    auto accessor = [&](size_t idx) {
        const Json& a = j["accessors"][idx];
        const Json& bv = j["bufferViews"][static_cast<size_t>(a["bufferView"].num)];
        Accessor r {};
        r.componentType = static_cast<int>(a["componentType"].num);
        r.components = ComponentCount(a["type"].str);
        r.count = static_cast<size_t>(a["count"].num);
        const size_t off = static_cast<size_t>(bv["byteOffset"].number(0)) + static_cast<size_t>(a["byteOffset"].number(0));
        const size_t bs = static_cast<size_t>(bv["byteStride"].number(0));
        r.stride = bs ? bs : static_cast<size_t>(ComponentSize(r.componentType) * r.components);
        r.base = reinterpret_cast<const uint8_t*>(buffers.at(static_cast<size_t>(bv["buffer"].num)).data()) + off;
        return r;
    };

    auto model = std::make_shared<Model>();
    auto loadTexture = [&](const Json& info) -> ResourceHandle<Image> {
        if (info.type != Json::Object) return ResourceHandle<Image>::Null();
        const Json& tex = j["textures"][static_cast<size_t>(info["index"].num)];
        const std::string uri = j["images"][static_cast<size_t>(tex["source"].num)]["uri"].str;
        // LoadTexture never fills its cache (source/model_loader.cpp:43-55): every reference loads again
        ImageCreation creation {};
        creation.name = uri;
        ResourceHandle<Image> handle {};
        if (LoadImageRGBA8(directory + "/" + uri, creation.data, creation.width, creation.height)) handle = _bindlessResources->Images().Create(creation);
        else std::fprintf(stderr, "[GLTF] Failed to load data from image [%s] from path [%s/%s]\n", uri.c_str(), directory.c_str(), uri.c_str());
        model->textures.push_back(handle);
        return handle;
    };

    for (const Json& gm : j["materials"].arr) {
        MaterialCreation mc {};
        const Json& pbr = gm["pbrMetallicRoughness"];
        mc.albedoMap = loadTexture(pbr["baseColorTexture"]);
        mc.metallicRoughnessMap = loadTexture(pbr["metallicRoughnessTexture"]);
        mc.normalMap = loadTexture(gm["normalTexture"]);
        // Assimp files glTF occlusion textures under aiTextureType_LIGHTMAP; the reference queries
        // aiTextureType_AMBIENT_OCCLUSION (source/model_loader.cpp:80) and finds none.
        mc.emissiveMap = loadTexture(gm["emissiveTexture"]);
        if (pbr.has("baseColorFactor")) for (size_t k = 0; k < 4; ++k) mc.albedoFactor[k] = static_cast<float>(pbr["baseColorFactor"][k].num);
        mc.metallicFactor = static_cast<float>(pbr["metallicFactor"].number(1.0));
        mc.roughnessFactor = static_cast<float>(pbr["roughnessFactor"].number(1.0));
        if (gm.has("normalTexture")) mc.normalScale = static_cast<float>(gm["normalTexture"]["scale"].number(1.0));
        if (gm.has("emissiveFactor")) for (size_t k = 0; k < 3; ++k) mc.emissiveFactor[k] = static_cast<float>(gm["emissiveFactor"][k].num);
        if (gm.has("occlusionTexture")) mc.occlusionStrength = static_cast<float>(gm["occlusionTexture"]["strength"].number(1.0));
        model->materials.push_back(_bindlessResources->Materials().Create(mc));
    }

    std::vector<std::vector<uint32_t>> meshesOfGltfMesh;
    for (const Json& gmesh : j["meshes"].arr) {
        std::vector<uint32_t> ids;
        for (const Json& prim : gmesh["primitives"].arr) {
            if (static_cast<int>(prim["mode"].number(4)) != 4) throw std::runtime_error("only triangle lists are imported");
            const Json& attr = prim["attributes"];
            Mesh mesh {};
            mesh.firstIndex = static_cast<uint32_t>(model->hostIndices.size());
            mesh.firstVertex = static_cast<uint32_t>(model->hostVertices.size());
            const Accessor pos = accessor(static_cast<size_t>(attr["POSITION"].num));
            const size_t nv = pos.count;
            model->hostVertices.resize(model->hostVertices.size() + nv);
            Model::Vertex* v = &model->hostVertices[mesh.firstVertex];
            for (size_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) v[i].position[c] = pos.getf(i, c);
            if (attr.has("NORMAL")) {
                const Accessor nrm = accessor(static_cast<size_t>(attr["NORMAL"].num));
                for (size_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) v[i].normal[c] = nrm.getf(i, c);
            }
            if (attr.has("TEXCOORD_0")) {
                const Accessor uv = accessor(static_cast<size_t>(attr["TEXCOORD_0"].num));
                for (size_t i = 0; i < nv; ++i) {
                    v[i].texCoord[0] = uv.getf(i, 0);
                    v[i].texCoord[1] = 1.0f - (1.0f - uv.getf(i, 1));   // importer flip, then aiProcess_FlipUVs
                }
            }
            if (prim.has("indices")) {
                const Accessor idx = accessor(static_cast<size_t>(prim["indices"].num));
                mesh.indexCount = static_cast<uint32_t>(idx.count);
                for (size_t i = 0; i < idx.count; ++i) model->hostIndices.push_back(mesh.firstVertex + static_cast<uint32_t>(idx.get(i, 0)));
            } else {
                mesh.indexCount = static_cast<uint32_t>(nv);
                for (size_t i = 0; i < nv; ++i) model->hostIndices.push_back(mesh.firstVertex + static_cast<uint32_t>(i));
            }
            if (!attr.has("NORMAL")) {   // aiProcess_GenNormals stand-in: flat face normals
                for (uint32_t t = 0; t + 2 < mesh.indexCount; t += 3) {
                    Model::Vertex* a = &model->hostVertices[model->hostIndices[mesh.firstIndex + t]];
                    Model::Vertex* b = &model->hostVertices[model->hostIndices[mesh.firstIndex + t + 1]];
                    Model::Vertex* c = &model->hostVertices[model->hostIndices[mesh.firstIndex + t + 2]];
                    double e1[3], e2[3];
                    for (int k = 0; k < 3; ++k) { e1[k] = double(b->position[k]) - a->position[k]; e2[k] = double(c->position[k]) - a->position[k]; }
                    double n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                    double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    if (l <= 0) l = 1;
                    for (Model::Vertex* q : { a, b, c }) for (int k = 0; k < 3; ++k) q->normal[k] = static_cast<float>(n[k] / l);
                }
            }
            if (prim.has("material")) {
                const size_t mi = static_cast<size_t>(prim["material"].num);
                if (mi < model->materials.size()) mesh.material = model->materials[mi];
            }
            ids.push_back(static_cast<uint32_t>(model->meshes.size()));
            model->meshes.push_back(mesh);
        }
        meshesOfGltfMesh.push_back(ids);
    }
    Upload(*model);

    // nodes: depth-first pre-order; the vector is reserved up front so parent pointers stay valid
    const Json& gnodes = j["nodes"];
    const Json& roots = j["scenes"][static_cast<size_t>(j["scene"].number(0))]["nodes"];
    model->nodes.reserve(gnodes.size() + 1);
    std::function<void(size_t, const Node*)> visit = [&](size_t gi, const Node* parent) {
        const Json& gn = gnodes[gi];
        Node& node = model->nodes.emplace_back();
        node.name = gn["name"].str;
        node.parent = parent;
        node.localMatrix = NodeMatrix(gn);
        if (gn.has("mesh")) node.meshes = meshesOfGltfMesh.at(static_cast<size_t>(gn["mesh"].num));
        const Node* self = &node;
        for (const Json& c : gn["children"].arr) visit(static_cast<size_t>(c.num), self);
    };
    if (roots.size() == 1) visit(static_cast<size_t>(roots[size_t(0)].num), nullptr);
    else {
        Node& root = model->nodes.emplace_back();
        root.name = "ROOT";
        const Node* self = &root;
        for (const Json& r : roots.arr) visit(static_cast<size_t>(r.num), self);
    }
    return model;
}

// ---------------------------------------------------------------------------------------- .ptscene
// little-endian container: "PTSC1\0\0\0", images, materials, models (see pt_amd/scene.py: save_ptscene)
std::shared_ptr<Model> ModelLoader::LoadPtScene(const std::string& path)
{
    const std::string data = ReadFile(path);
    size_t p = 0;
    auto take = [&](void* dst, size_t n) { if (p + n > data.size()) throw std::runtime_error("truncated .ptscene"); std::memcpy(dst, data.data() + p, n); p += n; };
    auto u32 = [&]() { uint32_t v; take(&v, 4); return v; };
    char magic[8];
    take(magic, 8);
    if (std::memcmp(magic, "PTSC1\0\0\0", 8) != 0) throw std::runtime_error("not a .ptscene file");
    auto model = std::make_shared<Model>();
    // images: index 0 of the file is the fallback the tables already hold
    const uint32_t nImages = u32();
    std::vector<uint32_t> imageMap(nImages, 0);
    for (uint32_t i = 0; i < nImages; ++i) {
        ImageCreation c {};
        c.width = u32(); c.height = u32();
        c.data.resize(size_t(c.width) * c.height * 4);
        take(c.data.data(), c.data.size());
        if (i == 0) continue;
        imageMap[i] = _bindlessResources->Images().Create(c).handle;
        model->textures.push_back(ResourceHandle<Image> { imageMap[i] });
    }
    const uint32_t nMaterials = u32();
    std::vector<uint32_t> materialMap(nMaterials);
    for (uint32_t i = 0; i < nMaterials; ++i) {
        PtMaterial m;
        take(&m, sizeof m);
        uint32_t* maps[5] = { &m.albedoMapIndex, &m.mrMapIndex, &m.normalMapIndex, &m.occlusionMapIndex, &m.emissiveMapIndex };
        for (uint32_t* q : maps) if (*q != NULL_RESOURCE_INDEX_VALUE && *q < nImages) *q = imageMap[*q];
        materialMap[i] = _bindlessResources->Materials().CreateRaw(m).handle;
        model->materials.push_back(ResourceHandle<Material> { materialMap[i] });
    }
    const uint32_t nModels = u32();
    if (nModels != 1) throw std::runtime_error("a .ptscene holds exactly one model");
    const uint32_t nv = u32();
    model->hostVertices.resize(nv);
    take(model->hostVertices.data(), sizeof(Model::Vertex) * nv);
    const uint32_t nidx = u32();
    model->hostIndices.resize(nidx);
    take(model->hostIndices.data(), 4 * size_t(nidx));
    const uint32_t nMeshes = u32();
    for (uint32_t i = 0; i < nMeshes; ++i) {
        Mesh m {};
        m.indexCount = u32(); m.firstIndex = u32(); m.firstVertex = u32();
        const uint32_t mat = u32();
        m.material.handle = mat < nMaterials ? materialMap[mat] : NULL_RESOURCE_INDEX_VALUE;
        model->meshes.push_back(m);
    }
    const uint32_t nNodes = u32();
    model->nodes.resize(nNodes);
    for (uint32_t i = 0; i < nNodes; ++i) {
        Node& n = model->nodes[i];
        int32_t parent; take(&parent, 4);
        n.parent = parent < 0 ? nullptr : &model->nodes[static_cast<size_t>(parent)];
        take(n.localMatrix.m, 64);
        const uint32_t nm = u32();
        n.meshes.resize(nm);
        take(n.meshes.data(), 4 * size_t(nm));
        const uint32_t nl = u32();
        n.name.resize(nl);
        take(n.name.data(), nl);
    }
    Upload(*model);
    return model;
}

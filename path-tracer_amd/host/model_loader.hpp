// model_loader.hpp -- mirrors include/model_loader.hpp:16-71 of the reference: Node, Mesh, Model and
// ModelLoader::LoadFromFile.  The reference imports through Assimp (source/model_loader.cpp:277);
// this loader restates the conventions Assimp's glTF 2.0 importer gives it (one mesh per primitive,
// depth-first node order, synthetic root for several scene roots, T*R*S node matrices in binary32, the
// double V flip) and additionally reads ".ptscene" containers written by pt_amd.scene.Scene.save_ptscene.
#pragma once
#include "resources.hpp"
#include <string_view>
#include <unordered_map>

struct Mat4 {              // row-major maths convention: m[row][col]
    float m[4][4];
    static Mat4 Identity();
};
Mat4 operator*(const Mat4& a, const Mat4& b);   // binary32, k accumulated 0..3 in order (glm's order)

struct Node {
    std::string name {};
    const Node* parent = nullptr;
    Mat4 localMatrix = Mat4::Identity();
    std::vector<uint32_t> meshes {};
    [[nodiscard]] Mat4 GetWorldMatrix() const;   // source/model_loader.cpp:253-265
};

struct Mesh {
    uint32_t indexCount {};
    uint32_t firstIndex {};
    uint32_t firstVertex {};
    ResourceHandle<Material> material {};
};

struct Model {
    struct Vertex {
        float position[3] {};
        float normal[3] {};
        float texCoord[2] {};
    };
    std::unique_ptr<Buffer> vertexBuffer;
    std::unique_ptr<Buffer> indexBuffer;
    uint32_t verticesCount {};
    uint32_t indexCount {};
    std::vector<Node> nodes {};
    std::vector<Mesh> meshes {};
    std::vector<ResourceHandle<Image>> textures {};
    std::vector<ResourceHandle<Material>> materials {};
    // host copies (the reference keeps none; used by tests and tools)
    std::vector<Vertex> hostVertices {};
    std::vector<uint32_t> hostIndices {};
};
static_assert(sizeof(Model::Vertex) == 32, "Vertex must match the device layout");

bool LoadPngRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height);
bool LoadJpegRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height);
// stbi_load(path, &w, &h, &channels, 4) of source/model_loader.cpp:25: PNG or JPEG by signature, RGBA8, top row first
bool LoadImageRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height);

class ModelLoader {
public:
    ModelLoader(const std::shared_ptr<BindlessResources>& bindlessResources, const std::shared_ptr<HipContext>& context);
    ~ModelLoader() = default;
    ModelLoader(const ModelLoader&) = delete;
    ModelLoader& operator=(const ModelLoader&) = delete;

    [[nodiscard]] std::shared_ptr<Model> LoadFromFile(std::string_view path);

private:
    [[nodiscard]] std::shared_ptr<Model> LoadGltf(const std::string& path);
    [[nodiscard]] std::shared_ptr<Model> LoadPtScene(const std::string& path);
    void Upload(Model& model);
    std::unordered_map<std::string, ResourceHandle<Image>> _imageCache {};
    std::shared_ptr<HipContext> _context;
    std::shared_ptr<BindlessResources> _bindlessResources;
};

// acceleration_structure.hpp -- mirrors include/acceleration_structure.hpp:7-14,
// include/bottom_level_acceleration_structure.hpp:12-37 and
// include/top_level_acceleration_structure.hpp:9-23 of the reference over the C ABI.
#pragma once
#include "model_loader.hpp"

struct BLASInput {
    Mat4 transform = Mat4::Identity();      // node world matrix (source/renderer.cpp:420)
    GeometryNodeCreation node {};
    // VkAccelerationStructureGeometryTrianglesDataKHR, source/renderer.cpp:427-441
    uint64_t vertexData = 0;                // device address of the model's vertex buffer
    uint32_t vertexStride = sizeof(Model::Vertex);
    uint32_t maxVertex = 0;
    uint64_t indexData = 0;                 // device address of the mesh's first index
    uint32_t primitiveCount = 0;
};

class BottomLevelAccelerationStructure {
public:
    BottomLevelAccelerationStructure(const BLASInput& input, const std::shared_ptr<BindlessResources>& resources, const std::shared_ptr<HipContext>& context);
    ~BottomLevelAccelerationStructure();
    BottomLevelAccelerationStructure(BottomLevelAccelerationStructure&& other) noexcept;
    BottomLevelAccelerationStructure& operator=(BottomLevelAccelerationStructure&& other) = delete;
    BottomLevelAccelerationStructure(const BottomLevelAccelerationStructure&) = delete;
    BottomLevelAccelerationStructure& operator=(const BottomLevelAccelerationStructure&) = delete;

    [[nodiscard]] PtBlas* Structure() const { return _structure; }
    [[nodiscard]] const Mat4& Transform() const { return _transform; }

private:
    PtBlas* _structure = nullptr;
    Mat4 _transform;
    std::shared_ptr<HipContext> _context;
};

class TopLevelAccelerationStructure {
public:
    TopLevelAccelerationStructure(const std::vector<BottomLevelAccelerationStructure>& blases, const std::shared_ptr<BindlessResources>& resources, const std::shared_ptr<HipContext>& context);
    ~TopLevelAccelerationStructure();
    TopLevelAccelerationStructure(const TopLevelAccelerationStructure&) = delete;
    TopLevelAccelerationStructure& operator=(const TopLevelAccelerationStructure&) = delete;

    [[nodiscard]] PtTlas* Structure() const { return _structure; }

private:
    PtTlas* _structure = nullptr;
    std::shared_ptr<HipContext> _context;
};

// acceleration_structure.cpp -- source/bottom_level_acceleration_structure.cpp:9-78 and
// source/top_level_acceleration_structure.cpp:8-113 restated over pt_blas_build / pt_tlas_build.
#include "acceleration_structure.hpp"

BottomLevelAccelerationStructure::BottomLevelAccelerationStructure(const BLASInput& input, const std::shared_ptr<BindlessResources>& resources, const std::shared_ptr<HipContext>& context)
    : _transform(input.transform), _context(context)
{
    PtCheckResult(pt_blas_build(context->Handle(), reinterpret_cast<const void*>(input.vertexData), input.vertexStride, input.maxVertex,
                                reinterpret_cast<const uint32_t*>(input.indexData), input.primitiveCount, 0, &_structure),
                  "Failed building bottom level acceleration structure!");
    resources->GeometryNodes().Create(input.node);
}

BottomLevelAccelerationStructure::~BottomLevelAccelerationStructure() { pt_blas_destroy(_structure); }

BottomLevelAccelerationStructure::BottomLevelAccelerationStructure(BottomLevelAccelerationStructure&& other) noexcept
    : _structure(other._structure), _transform(other._transform), _context(other._context)
{
    other._structure = nullptr;
}

// VkGLMToTransformMatrixKHR (source/vk_common.cpp:205-212): the 3x4 row-major upper part
static void ToTransformMatrix(const Mat4& m, float out[12])
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = m.m[r][c];
}

TopLevelAccelerationStructure::TopLevelAccelerationStructure(const std::vector<BottomLevelAccelerationStructure>& blases, const std::shared_ptr<BindlessResources>& resources, const std::shared_ptr<HipContext>& context)
    : _context(context)
{
    uint32_t geometryCount = 0;
    std::vector<PtInstance> instances {};
    for (const auto& blas : blases) {
        PtInstance& instance = instances.emplace_back();
        ToTransformMatrix(blas.Transform(), instance.transform);
        const uint32_t customIndex = static_cast<uint32_t>(instances.size() - 1);
        instance.instanceCustomIndexAndMask = (customIndex & 0xFFFFFFu) | (0xFFu << 24);
        instance.sbtOffsetAndFlags = 0u | (0x1u << 24);   // VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR
        instance.accelerationStructureReference = reinterpret_cast<uint64_t>(blas.Structure());
        BLASInstanceCreation blasInstanceCreation {};
        blasInstanceCreation.firstGeometryIndex = geometryCount;
        resources->BLASInstances().Create(blasInstanceCreation);
        geometryCount++;
    }
    PtCheckResult(pt_tlas_build(context->Handle(), instances.data(), static_cast<uint32_t>(instances.size()), &_structure),
                  "Failed building top level acceleration structure!");
}

TopLevelAccelerationStructure::~TopLevelAccelerationStructure() { pt_tlas_destroy(_structure); }

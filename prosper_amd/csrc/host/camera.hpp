// host/camera.hpp — scene::Camera of the headless host layer.
//
// Restates the parts of prosper's camera the RT reference pass consumes
// (reference: src/scene/Camera.hpp:22-48, src/scene/Camera.cpp:105-204,366-395): a right-handed
// look-at worldToCamera, a reverse-Z, Y-flipped perspective cameraToClip, the CameraUniforms
// block and CameraParameters::focalLength.  TAA jitter, frustum planes for culling and the
// gesture offsets of the interactive app are raster/UI features and stay out of scope.
#pragma once

#include <cstdint>

#include "../../../include/prosper_pt/shader_structs.h"

namespace scene
{

struct CameraTransform
{
    float eye[3] = {1.f, 0.5f, 1.f};
    float target[3] = {0.f, 0.f, 0.f};
    float up[3] = {0.f, 1.f, 0.f};
};

struct CameraParameters
{
    float fov = 59.f * 3.14159265358979323846f / 180.f; // glm::radians(59.f)
    float zN = 0.1f;
    float zF = 100.f;
    float apertureDiameter = 0.00001f;
    float focusDistance = 1.f;
    float focalLength = 0.f;
};

class Camera
{
  public:
    void lookAt(const CameraTransform &transform);
    void setParameters(const CameraParameters &parameters);
    void updateResolution(uint32_t width, uint32_t height);
    // Camera::updateBuffer: recomputes the projection and returns the uniforms block
    const prosper_CameraUniforms &updateBuffer();
    // cleared by endFrame(); feeds ReferencePC skipHistory (RtReference.cpp:282-283)
    [[nodiscard]] bool changedThisFrame() const { return m_changedThisFrame; }
    void endFrame() { m_changedThisFrame = false; }
    [[nodiscard]] const CameraParameters &parameters() const { return m_parameters; }
    [[nodiscard]] const prosper_CameraUniforms &uniforms() const { return m_uniforms; }
    [[nodiscard]] static float sensorWidth() { return 0.035f; }

  private:
    void updateWorldToCamera();
    void perspective();

    CameraTransform m_transform;
    CameraParameters m_parameters;
    uint32_t m_resolution[2] = {1, 1};
    float m_worldToCamera[16] = {};
    float m_cameraToWorld[16] = {};
    float m_cameraToClip[16] = {};
    float m_clipToWorld[16] = {};
    float m_maxViewScale = 1.f;
    bool m_changedThisFrame = true;
    prosper_CameraUniforms m_uniforms = {};
};

} // namespace scene
